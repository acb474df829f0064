"""Run-to-run identity of the shipped config's train step: two in-line runners and one with the prefetched front, same
weights, batches and SRL seeds; losses of 4 steps and the first parameter whose values differ after each step."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic
from gga_amd.train import Runner
DEV = 'cuda:0'
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
cfg = Config.fromfile(os.path.join(root, 'configs', 'gga', sys.argv[1] if len(sys.argv) > 1 else 'gga_kitti_config.py'))
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV)
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for name in ('reg', 'height', 'dim', 'rot'):
            getattr(th, name)[-1].weight.mul_(0.05)
rng = synthetic.RANGE_SECOND if 'pointpillars' not in cfg.filename else synthetic.RANGE_PP
batches = []
for i in range(2):
    b = synthetic.make_batch(2, start=10 * i, n_points=8000, pc_range=rng)
    b['points'] = [p.to(DEV) for p in b['points']]
    batches.append({k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)})
models = [model, copy.deepcopy(model), copy.deepcopy(model)]
runners = [Runner(m, cfg, max_iters=100) for m in models]
names = ['inline-1', 'inline-2', 'prefetch']
for i in range(4):
    ls = []
    for r, n in zip(runners, names):
        torch.manual_seed(100 + i)
        out = r.step(batches[i % 2], next_data=batches[(i + 1) % 2] if n == 'prefetch' else None)
        torch.cuda.synchronize()
        ls.append(float(out['loss']))
    print(f'step {i}: ' + '  '.join(f'{n} {l!r}' for n, l in zip(names, ls)))
    for other in (1, 2):
        for (pn, p), (_, q) in zip(models[0].named_parameters(), models[other].named_parameters()):
            if not torch.equal(p, q):
                print(f'   after step {i}: {names[other]} first differs from inline-1 at {pn} (max |diff| {float((p - q).abs().max()):.3e}, rel {float((p - q).norm() / p.norm()):.3e})')
                break
