"""Which objects of a train step die only in the cyclic garbage collector (reference cycles), and how much device memory do
they hold until it runs? (round 5: the sparse leg's caching allocator kept calling hipMalloc - +1.6 GB reserved per step - because
the step's tensors were not freed by reference counting.) Steps a config a few times with the collector off, then collects
with DEBUG_SAVEALL and prints: garbage objects by type, the device tensors among them (shape, MB), and for the largest ones the
types of the garbage objects that refer to them. Usage: find_cycles.py pp|second [steps]"""
import gc
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
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
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
for i in range(4):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
gc.collect()
gc.freeze()
gc.disable()
m0 = torch.cuda.memory_stats(dev)
for i in range(steps):
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
m1 = torch.cuda.memory_stats(dev)
print(f'{which}: {steps} steps with the collector off: device mallocs +{m1["num_device_alloc"] - m0["num_device_alloc"]}, reserved '
      f'{m0["reserved_bytes.all.current"] / 2**20:.0f} -> {m1["reserved_bytes.all.current"] / 2**20:.0f} MB, active '
      f'{m0["active_bytes.all.current"] / 2**20:.0f} -> {m1["active_bytes.all.current"] / 2**20:.0f} MB')
gc.set_debug(gc.DEBUG_SAVEALL)
n = gc.collect()
garbage = list(gc.garbage)
gc.set_debug(0)
print(f'collected {n} objects; by type:')
for t, c in Counter(type(o).__module__ + '.' + type(o).__qualname__ for o in garbage).most_common(25):
    print(f'  {c:6d} {t}')
ids = {id(o): o for o in garbage}
tensors = [o for o in garbage if isinstance(o, torch.Tensor) and o.is_cuda]
seen, total = set(), 0
for t in tensors:
    p = t.untyped_storage().data_ptr()
    if p not in seen:
        seen.add(p)
        total += t.untyped_storage().nbytes()
print(f'device tensors in the garbage: {len(tensors)} ({total / 2**20:.0f} MB of distinct storage, {total / 2**20 / steps:.0f} MB per step)')
tensors.sort(key=lambda t: -t.untyped_storage().nbytes())
for t in tensors[:12]:
    refs = [r for r in gc.get_referrers(t) if id(r) in ids]
    kinds = Counter(type(r).__module__ + '.' + type(r).__qualname__ for r in refs)
    detail = []
    for r in refs[:4]:
        if isinstance(r, dict):
            owners = [type(o).__qualname__ for o in gc.get_referrers(r) if id(o) in ids and not isinstance(o, (list, dict))][:3]
            detail.append('dict keys ' + str([k for k, v in r.items() if v is t][:3]) + ' of ' + str(owners))
        elif isinstance(r, (tuple, list)):
            owners = [type(o).__qualname__ for o in gc.get_referrers(r) if id(o) in ids][:3]
            detail.append(type(r).__name__ + ' of ' + str(owners))
        else:
            detail.append(type(r).__qualname__)
    print(f'  {tuple(t.shape)} {t.dtype} {t.untyped_storage().nbytes() / 2**20:.1f} MB  grad_fn {type(t.grad_fn).__name__ if t.grad_fn is not None else None}  <- {dict(kinds)} | {detail}')
