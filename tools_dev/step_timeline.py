"""Where does the wall time of a free-running train loop go on the host? Per step of the bench loop (no synchronisation
between steps, as bench.py times it): host time inside the prefetch of the next batch's front (its count reads wait for the
side stream), host time of the rest of the step (queueing), and the step-to-step period; afterwards the totals against the
wall time and the device-side time of the same steps (events at the start and end of every step on the main stream).
Usage: step_timeline.py [second|pp] [steps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
import torch
import gga_amd  # noqa: F401
from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner, setup_multi_processes
import bench

dev = torch.device('cuda:0')
SECOND = len(sys.argv) > 1 and sys.argv[1] == 'second'
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = Config.fromfile(bench.SECOND_CONFIG if SECOND else bench.PP_CONFIG)
setup_multi_processes(cfg)
cfg.model.pts_middle_encoder['channels_last'] = True
BS = 8 if SECOND else 16
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(dev))
bench.damp_head_init(model, 0.05)
model.train()
runner = Runner(model, cfg, max_iters=1000, distributed=False, device=dev)
pc_range = tuple(cfg.model.pts_voxel_layer.point_cloud_range)
batches = []
for i in range(2):
    b = synthetic.make_batch(BS, start=i * BS, rank=0, pc_range=pc_range)
    b['points'] = [p.to(dev) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
torch.cuda.synchronize()
runner.inputs_ready(*batches)
pf = []
real_prefetch = runner.prefetch


def timed_prefetch(data):
    t = time.perf_counter()
    real_prefetch(data)
    pf.append(time.perf_counter() - t)


runner.prefetch = timed_prefetch
for i in range(5):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
runner.freeze_gc()
del pf[:]
starts, hosts, evs = [], [], []
t_all = time.perf_counter()
for i in range(N):
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    t0 = time.perf_counter()
    runner.step(batches[(5 + i) % 2], next_data=batches[(5 + i + 1) % 2])      # (the alternation continues across the warm-up)
    hosts.append(time.perf_counter() - t0)
    starts.append(t0)
    evs.append(e0)
e_end = torch.cuda.Event(enable_timing=True)
e_end.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t_all) / N * 1e3
dev_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(N - 1)] + [evs[-1].elapsed_time(e_end)]
pfm = np.array(pf[:N]) * 1e3 if pf else np.zeros(N)
hm = np.array(hosts) * 1e3
period = np.diff(np.array(starts)) * 1e3
q = lambda a: f'median {np.median(a):6.2f}  p90 {np.percentile(a, 90):6.2f}  max {np.max(a):6.2f}'
print(f'{"second" if SECOND else "pp"}: wall {wall:.2f} ms/step over {N} steps; load average {os.getloadavg()}; cpus {os.cpu_count()}')
print('host per step        ', q(hm))
print('  of it: prefetch    ', q(pfm))
print('  of it: queueing    ', q(hm - pfm))
print('host step period     ', q(period))
print('device step-to-step  ', q(np.array(dev_ms)))
print('device ms per step   ', ' '.join(f'{d:.1f}' for d in dev_ms))
print('host ms per step     ', ' '.join(f'{d:.1f}' for d in hm))
print('prefetch ms per step ', ' '.join(f'{d:.1f}' for d in pfm))
