"""Where does the train thread spend (or lose) its time inside a step once the loop is in steady state? Wall-clock per phase of
Runner.step on the host (no device synchronisation added): extract_feat, head forward, pack_targets, the loss launches, backward,
clipping, optimizer - a phase that takes far longer than its first-steps figure is where the host waits for the device.
Usage: host_phases.py [second|pp]"""
import os, sys, time, collections
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
T = collections.defaultdict(list)


def wrap(obj, name, label):
    f = getattr(obj, name)

    def g(*a, **k):
        t = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            T[label].append(time.perf_counter() - t)
    setattr(obj, name, g)


head = model.pts_bbox_head
wrap(model, 'extract_feat', 'extract_feat')
wrap(head, 'forward', 'head.forward')
wrap(head, 'pack_targets', 'head.pack_targets')
wrap(head, 'loss', 'head.loss (incl. pack_targets)')
wrap(runner.optimizer, 'step', 'optimizer.step')
wrap(runner.optimizer, 'zero_grad', 'zero_grad')
wrap(runner, '_bound_lead', '_bound_lead')
_bw = torch.Tensor.backward


def bw(self, *a, **k):
    t = time.perf_counter()
    try:
        return _bw(self, *a, **k)
    finally:
        T['backward'].append(time.perf_counter() - t)


torch.Tensor.backward = bw
_clip = torch.nn.utils.clip_grad_norm_


def clip(*a, **k):
    t = time.perf_counter()
    try:
        return _clip(*a, **k)
    finally:
        T['clip_grad_norm_'].append(time.perf_counter() - t)


torch.nn.utils.clip_grad_norm_ = clip
N = 40
for i in range(N):
    t = time.perf_counter()
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
    T['step'].append(time.perf_counter() - t)
torch.cuda.synchronize()
print(f'{"second" if SECOND else "pp"}: host milliseconds per phase, median of steps 0-4 | median of steps 15-39')
for k, v in T.items():
    v = np.array(v) * 1e3
    per = len(v) // N
    v = v[:per * N].reshape(N, per).sum(1) if per else v
    print(f'{k:34s} {np.median(v[:5]):8.2f} | {np.median(v[15:]):8.2f}')
