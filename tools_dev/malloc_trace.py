"""Who calls hipMalloc in steady state? Steps gga_kitti_config.py (bs 8) 6 times, then records the caching allocator's history
for 4 more steps and prints every segment allocation (= hipMalloc) of that window: size, stream, and the innermost frames of
this repo on its Python stack. Usage: malloc_trace.py [second|pp]"""
import os
import sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from gga_amd import Config, build_model, synthetic
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner, setup_multi_processes

which = sys.argv[1] if len(sys.argv) > 1 else 'second'
dev = torch.device('cuda:0')
path, rng, B = (('gga_kitti_pointpillars_config.py', synthetic.RANGE_PP, 16) if which == 'pp' else ('gga_kitti_config.py', synthetic.RANGE_SECOND, 8))
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', path))
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
    b = synthetic.make_batch(B, start=B * i, pc_range=rng)
    b['points'] = [p.to(dev) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
runner.inputs_ready(*batches)
for i in range(6):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.memory._record_memory_history(max_entries=200000)
m0 = torch.cuda.memory_stats(dev)
for i in range(6, 10):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
m1 = torch.cuda.memory_stats(dev)
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
print(f'{which}: 4 steps: device mallocs +{m1["num_device_alloc"] - m0["num_device_alloc"]}, frees +{m1["num_device_free"] - m0["num_device_free"]}, '
      f'reserved {m0["reserved_bytes.all.current"] / 2**20:.0f} -> {m1["reserved_bytes.all.current"] / 2**20:.0f} MB')
main = torch.cuda.current_stream(dev).cuda_stream
rows = Counter()
for trace in snap['device_traces']:
    for ev in trace:
        if ev['action'] != 'segment_alloc':
            continue
        frames = [f for f in ev.get('frames', []) if REPO in f['filename'] and 'tools_dev' not in f['filename']]
        where = ' < '.join(f"{os.path.basename(f['filename'])}:{f['line']} {f['name']}" for f in frames[:4])
        rows[(ev['size'] >> 20, 'main' if ev['stream'] == main else 'side %x' % ev['stream'], where)] += 1
for (mb, stream, where), c in sorted(rows.items(), key=lambda kv: -kv[0][0] * kv[1])[:40]:
    print(f'{c:3d} x {mb:6d} MB  {stream:14s} {where}')
# the live segments by stream
segs = Counter()
for s in snap['segments']:
    segs[('main' if s['stream'] == main else 'side %x' % s['stream'], s['segment_type'])] += s['total_size'] >> 20
print('segments (MB) by stream / pool:', dict(segs))
