"""After N steps of gga_kitti_config.py: which PreparedInputs / _Level objects are alive and who refers to them?"""
import gc
import os
import sys
from collections import Counter

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from gga_amd import Config, build_model, synthetic, detectors, sparse
from gga_amd.cnn import to_channels_last
from gga_amd.train import Runner, setup_multi_processes

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device('cuda:0')
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
setup_multi_processes(cfg)
cfg.model.pts_middle_encoder['channels_last'] = True
torch.manual_seed(0)
model = to_channels_last(build_model(cfg.model).to(dev)).train()
runner = Runner(model, cfg, max_iters=1000, device=dev)
batches = []
for i in range(2):
    b = synthetic.make_batch(8, start=8 * i, pc_range=synthetic.RANGE_SECOND)
    b['points'] = [p.to(dev) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
runner.inputs_ready(*batches)
for i in range(steps):
    if i == 3:
        runner.freeze_gc()
    runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
torch.cuda.synchronize()
print('retired', len(runner._retired), 'prepared', len(runner._prepared), 'in flight', len(runner._in_flight),
      'active MB', torch.cuda.memory_stats(dev)['active_bytes.all.current'] >> 20)
objs = gc.get_objects() + (gc.get_freeze_count() and [])
preps = [o for o in gc.get_objects() if isinstance(o, detectors.PreparedInputs)]
levels = [o for o in gc.get_objects() if isinstance(o, sparse._Level)]
print('PreparedInputs alive:', len(preps), ' _Level alive:', len(levels), ' IndexPlan alive:', sum(isinstance(o, sparse.IndexPlan) for o in gc.get_objects()))



def owner_of(d):
    for o in gc.get_referrers(d):
        if getattr(o, '__dict__', None) is d:
            return o
    return None


def describe(o):
    t = type(o)
    s = t.__module__ + '.' + t.__qualname__
    if isinstance(o, dict):
        own = owner_of(o)
        s += (' (__dict__ of %s)' % (type(own).__module__ + '.' + type(own).__qualname__) if own is not None else '') + ' keys=' + str(list(o.keys())[:5])
    if isinstance(o, (list, tuple)):
        s += f' len={len(o)} of ' + str(sorted({type(x).__name__ for x in o})[:4])
    if isinstance(o, torch.Tensor):
        s += f' shape={tuple(o.shape)} grad_fn={type(o.grad_fn).__name__ if o.grad_fn is not None else None}'
    return s


skip = set()


def chain(o, depth=0, seen=None, limit=5):
    seen = seen if seen is not None else set()
    if depth > limit or id(o) in seen:
        return
    seen.add(id(o))
    for r in gc.get_referrers(o):
        if id(r) in skip or isinstance(r, type(sys._getframe())) or type(r).__name__ in ('list_iterator', 'cell'):
            continue
        print('  ' * (depth + 1) + '<- ' + describe(r)[:200])
        if not isinstance(r, (type, type(sys))):
            chain(r, depth + 1, seen, limit)


plans = [o for o in gc.get_objects() if isinstance(o, sparse.IndexPlan)]
held = {id(p.coors) for p in preps} | {id(getattr(p, 'coors', None)) for p in preps}
skip |= {id(plans), id(preps), id(levels), id(objs)}
orphans = [pl for pl in plans if not any(any(v is pl for v in vars(p).values()) for p in preps)]
print('IndexPlans not owned by a live PreparedInputs:', len(orphans))
for pl in orphans[:2]:
    print('IndexPlan', hex(id(pl)))
    chain(pl, limit=6)
