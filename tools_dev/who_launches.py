"""Which Python lines of the package launch the SMALL kernels of a train step (fills, memsets, copies, cats)? One step of the
bench loop under torch.profiler with Python stacks; every device kernel shorter than 10 us is attributed to the
CPU op that launched it and the chain of ops that called it (custom autograd functions appear by name). Usage: who_launches.py [second|pp]"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import gga_amd  # noqa: F401
from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner, setup_multi_processes
import bench

dev = torch.device('cuda:0')
SECOND = len(sys.argv) > 1 and sys.argv[1] == 'second'
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
for i in range(6):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for i in range(2):
        runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
by_site = collections.Counter()
time_site = collections.Counter()
names = collections.defaultdict(collections.Counter)
for ev in prof.events():
    ks = [k for k in ev.kernels if k.duration < 10.0] if hasattr(ev, 'kernels') else []
    if not ks:
        continue
    chain, e = [], ev                      # the launching op and the ops it was called from (outermost last)
    while e is not None and len(chain) < 6:
        chain.append(e.name)
        e = e.cpu_parent
    site = ' < '.join(chain)
    for k in ks:
        by_site[site] += 1
        time_site[site] += k.duration
        names[site][k.name[:60]] += 1
tot = sum(by_site.values())
print(f'{"second" if SECOND else "pp"}: {tot / 2:.0f} kernels under 10 us per step, {sum(time_site.values()) / 2e3:.2f} ms per step')
for site, n in by_site.most_common(60):
    print(f'{n / 2:6.1f} per step {time_site[site] / 2:7.1f} us  {site[:110]}  <- {", ".join(f"{c // 2 if c > 1 else c}x {nm}" for nm, c in names[site].most_common(3))}')
