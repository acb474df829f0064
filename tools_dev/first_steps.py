"""what is the sparse leg's first-process transient? The first N steps of gga_kitti_config.py (bs 8) in a
fresh process, per step: device time (HIP events around the step on the main stream), host time of the step call, the caching
allocator's counters (device mallocs = hipMalloc calls, alloc retries, segments, reserved bytes) and the sclk / power the SMU
reports (sysfs). Usage: first_steps.py [steps] [label] -> one JSON line per step on stdout, a summary line at the end."""
import glob
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner, setup_multi_processes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
label = sys.argv[2] if len(sys.argv) > 2 else 'run'
presize = os.environ.get('GGA_FIRST_STEPS_WARM_ALLOC') == '1'
dev = torch.device('cuda:0')


def smu():
    out = {}
    for card in glob.glob('/sys/class/drm/card*/device'):
        try:
            sclk = [l for l in open(os.path.join(card, 'pp_dpm_sclk')).read().splitlines() if l.strip().endswith('*')]
            out['sclk'] = sclk[0].split(':')[1].strip(' *') if sclk else None
        except OSError:
            pass
        for hw in glob.glob(os.path.join(card, 'hwmon', 'hwmon*')):
            for name, key in (('power1_average', 'power_w'), ('power1_input', 'power_w'), ('temp1_input', 'temp_c')):
                try:
                    out[key] = int(open(os.path.join(hw, name)).read()) / (1e6 if 'power' in name else 1e3)
                except (OSError, ValueError):
                    pass
        if out:
            break
    return out


cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
setup_multi_processes(cfg)
cfg.model.pts_middle_encoder['channels_last'] = True
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(dev)).train()
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for n in ('reg', 'height', 'dim', 'rot'):
            getattr(th, n)[-1].weight.mul_(0.05)
runner = Runner(model, cfg, max_iters=1000, device=dev)
batches = []
for i in range(2):
    b = synthetic.make_batch(8, start=8 * i, pc_range=synthetic.RANGE_SECOND)
    b['points'] = [p.to(dev) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
torch.cuda.synchronize()
runner.inputs_ready(*batches)
rows = []
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
host = []
stats = []
for i in range(steps):
    if i == 3:
        runner.freeze_gc()
    ev[i][0].record()
    t = time.perf_counter()
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
    host.append((time.perf_counter() - t) * 1e3)
    ev[i][1].record()
    ms = torch.cuda.memory_stats(dev)
    stats.append(dict(device_mallocs=ms.get('num_device_alloc', 0), device_frees=ms.get('num_device_free', 0), retries=ms.get('num_alloc_retries', 0),
                      segments=ms.get('segment.all.current', 0), reserved_mb=round(ms.get('reserved_bytes.all.current', 0) / 2 ** 20),
                      active_mb=round(ms.get('active_bytes.all.current', 0) / 2 ** 20), **smu()))
torch.cuda.synchronize()
for i in range(steps):
    row = dict(label=label, step=i, device_ms=round(ev[i][0].elapsed_time(ev[i][1]), 2), host_ms=round(host[i], 2), **stats[i])
    print(json.dumps(row))
dms = [ev[i][0].elapsed_time(ev[i][1]) for i in range(steps)]
print(json.dumps(dict(label=label, summary=True, first5=round(sum(dms[:5]) / 5, 2), steps5_15=round(sum(dms[5:15]) / 10, 2),
                      last10=round(sum(dms[-10:]) / 10, 2), mallocs_after_step5=stats[-1]['device_mallocs'] - stats[5]['device_mallocs'])))
