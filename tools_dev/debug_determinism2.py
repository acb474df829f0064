"""Which parameter gradients differ between two runs of the SAME step (same weights, batch, SRL draws)?"""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, dense_conv
DEV = 'cuda:0'
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
cfg = Config.fromfile(os.path.join(root, 'configs', 'gga', sys.argv[1] if len(sys.argv) > 1 else 'gga_kitti_config.py'))
dense_conv.PLANES = int(sys.argv[2]) if len(sys.argv) > 2 else 2
if len(sys.argv) > 3:
    cfg.model.pts_middle_encoder['channels_last'] = True
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV)
if len(sys.argv) > 3:
    from gga_amd.cnn import to_channels_last
    model = to_channels_last(model)
with torch.no_grad():
    for th in model.pts_bbox_head.task_heads:
        for name in ('reg', 'height', 'dim', 'rot'):
            getattr(th, name)[-1].weight.mul_(0.05)
rng = synthetic.RANGE_SECOND if 'pointpillars' not in cfg.filename else synthetic.RANGE_PP
b = synthetic.make_batch(2, n_points=8000, pc_range=rng)
b['points'] = [p.to(DEV) for p in b['points']]
data = {k: b[k] for k in synthetic.BATCH_KEYS + ('img_metas',)}
grads = []
for rep in range(3):
    m = copy.deepcopy(model).train()
    dense_conv.AMAX_POOL.next_generation()
    torch.manual_seed(5)
    out = m.train_step(data)
    out['loss'].backward()
    torch.cuda.synchronize()
    grads.append(({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}, float(out['loss'])))
print('losses', [g[1] for g in grads])
for other in (1, 2):
    bad = [(n, float((grads[0][0][n] - g).abs().max() / (g.abs().max() + 1e-30))) for n, g in grads[other][0].items() if not torch.equal(grads[0][0][n], g)]
    print(f'run {other} vs run 0: {len(bad)} of {len(grads[0][0])} gradients differ')
    for n, e in bad[:40]:
        print(f'   {n:60s} {e:.2e}')
