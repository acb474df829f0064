"""Sparse rows of tests/test_precision_gpu.py only (the shipped config's sparse convolution shapes on real step operands),
plus the kernel time of each product, for the environment this process was started with
(GGA_SP_OFFSET_SUMS / GGA_SP_HALO / GGA_SP_HALO_CHAIN are read once per process).
    python tools_dev/sparse_precision.py TAG        -> gpurun_out/sparse_precision_TAG.json
"""
import json
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
import test_precision_gpu as P          # noqa: E402
from gga_amd import Config, build_model, synthetic, dense_conv     # noqa: E402
from gga_amd.cnn import to_channels_last       # noqa: E402
from gga_amd.train import Runner               # noqa: E402

tag = sys.argv[1] if len(sys.argv) > 1 else 'default'
B = int(os.environ.get('SP_PREC_FRAMES', '2'))
torch.set_num_threads(min(os.cpu_count() or 1, 32))
dense_conv.PLANES_PINNED = False
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
cfg.model.pts_middle_encoder['channels_last'] = True
torch.manual_seed(2)
model = to_channels_last(build_model(cfg.model).to('cuda:0')).train()
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for n in ('reg', 'height', 'dim', 'rot'):
            getattr(th, n)[-1].weight.mul_(0.05)
b = synthetic.make_batch(B, start=300, n_points=20000, pc_range=synthetic.RANGE_SECOND)
b['points'] = [p.to('cuda:0') for p in b['points']]
data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
runner = Runner(model, cfg, max_iters=100)
for _ in range(3):
    runner.step(data)
dense_conv.PLANES = 2
dense_conv.AMAX_POOL.next_generation()
records = P._capture(model, data)
rows = []
for key, rec in sorted(records.items(), key=lambda kv: str(kv[0])):
    if rec['kind'] != 'sparse':
        continue
    nbr = rec['rb'].nbr if hasattr(rec['rb'], 'nbr') else rec['rb']
    ref64 = list(P._sparse_ref(rec['x'], rec['w'], nbr, rec['gy'], torch.float64))
    ref32 = P._sparse_ref(rec['x'], rec['w'], nbr, rec['gy'], torch.float32)
    got = P._product(rec, 2)
    # time: the whole forward + backward of the product (three kernels + their small helpers), median of 5
    ts = []
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        P._product(rec, 2)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[1:])
    for di, direction in enumerate(('forward', 'backward_data', 'weight_gradient')):
        if got[di] is None or (direction == 'backward_data' and key[1] == 4):
            continue
        rms, mx = P._err(got[di].reshape(ref64[di].shape), ref64[di])
        rms32, _ = P._err(ref32[di].reshape(ref64[di].shape), ref64[di])
        rows.append(dict(shape='%s %d->%d k%s s%s @%d' % (key[0], key[1], key[2], 'x'.join(map(str, key[3])), 'x'.join(map(str, key[4])), key[5][0]),
                         direction=direction, planes2_rms=rms, sgemm_rms=rms32, ratio=rms / rms32, fwd_bwd_ms=ts[len(ts) // 2]))
        print('%-8s %-44s %-16s planes2 %.2e sgemm %.2e ratio %.2f  (fwd+bwd %.3f ms)' % (tag, rows[-1]['shape'], direction, rms, rms32, rms / rms32, ts[len(ts) // 2]), flush=True)
os.makedirs(os.path.join(REPO, 'gpurun_out'), exist_ok=True)
json.dump(dict(tag=tag, env={k: v for k, v in os.environ.items() if k.startswith('GGA_')}, rows=rows),
          open(os.path.join(REPO, 'gpurun_out', f'sparse_precision_{tag}.json'), 'w'), indent=1)
