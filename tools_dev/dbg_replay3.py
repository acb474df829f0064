"""Twins with a few fingerprints per step (prepared voxels / counts / coordinates / rule books, voxel features, BEV map, loss): at the
first step that differs, which of them differs first?"""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
import bench
from gga_amd import detectors
STEPS = int(os.environ.get('GGA_REPLAY_STEPS', '10'))
TWINS = int(os.environ.get('GGA_REPLAY_TWINS', '40'))
args = bench.parse_args(['--batch', '8', '--steps', '1', '--warmup', '0', '--no-cpu-baseline', '--no-roofline'])
torch.manual_seed(0)
run = bench.run_workload(bench.SECOND_CONFIG, 8, 1, 0, args, 0, 1, torch.device('cuda:0'))
runner, batches = run['runner'], run['batches']
model = runner.raw_model
state0 = copy.deepcopy(model.state_dict())
opt0 = copy.deepcopy(runner.optimizer.state_dict())
log = None
grad_names = []


_w = {}


def fp(t):
    """sum and sum of squares in float64 (order-insensitive up to the reduction's own fixed tree)"""
    t = t.detach().double().flatten()
    return torch.stack([t.sum(), (t * t).sum()])


orig_ft = type(model).forward_train


def ft(self, points=None, **kw):
    if isinstance(points, detectors.PreparedInputs):
        log.append(('prep voxels', fp(points.voxels)))
        log.append(('prep num_points', fp(points.num_points)))
        log.append(('prep coors', fp(points.coors)))
        plan = getattr(points.coors, 'index_plan', None) if os.environ.get('GGA_REPLAY_PLAN') == '1' else None
        if plan is not None:
            for i, t in enumerate(plan.tensors()):
                if t.dtype != torch.uint8:           # (the hash tables: their slot layout depends on which insert won a collision)
                    log.append((f'plan tensor {i} {tuple(t.shape)} {t.dtype}', fp(t)))
    return orig_ft(self, points=points, **kw)


type(model).forward_train = ft
model.pts_middle_encoder.register_forward_hook(lambda m, i, o: log.append(('bev map', fp(o))))
if os.environ.get('GGA_REPLAY_LAYERS', '1') == '1':
    enc = model.pts_middle_encoder
    enc.conv_input.register_forward_hook(lambda m, i, o: log.append(('conv_input out', fp(o.features))))
    for li, layer in enumerate(enc.encoder_layers):
        layer.register_forward_hook(lambda m, i, o, li=li: log.append((f'encoder layer {li} out ({o.features.shape[0]} rows)', fp(o.features))))
    enc.conv_out.register_forward_hook(lambda m, i, o: log.append(('conv_out out', fp(o.features))))
model.pts_voxel_encoder.register_forward_hook(lambda m, i, o: log.append(('voxel features', fp(o))))
model.pts_backbone.register_forward_hook(lambda m, i, o: log.append(('backbone out', fp(o[0] if isinstance(o, (list, tuple)) else o))))


def replay():
    global log
    model.load_state_dict(state0)
    runner.optimizer.load_state_dict(copy.deepcopy(opt0))
    runner.iter = 1
    runner._prepared.clear()
    torch.manual_seed(123)
    steps = []
    for i in range(STEPS):
        log = []
        out = runner.step(batches[i % 2], next_data=batches[(i + 1) % 2])
        log.append(('loss', fp(out['loss'])))
        if os.environ.get('GGA_REPLAY_GRADS', '1') == '1':      # one multi-tensor launch: the norm of every parameter gradient of this step
            gl = [(n_, p_.grad) for n_, p_ in model.named_parameters() if p_.grad is not None]
            norms = torch.stack(torch._foreach_norm([g for _, g in gl]))
            log.append(('grad norms', norms))
            grad_names[:] = [n_ for n_, _ in gl]
        for k, v in sorted(out['log_vars'].items()):
            log.append(('log ' + k, fp(torch.as_tensor(v))))
        steps.append(log)
    torch.cuda.synchronize()
    return steps


a = replay()
bad = 0
for rep in range(TWINS):
    b = replay()
    first = None
    for i, (sa, sb) in enumerate(zip(a, b)):
        for (na, va), (nb, vb) in zip(sa, sb):
            if na != nb or not torch.equal(va, vb):
                if na == 'grad norms':
                    idx = (va != vb).nonzero().flatten().tolist()
                    rel = ((va - vb).abs() / va.abs().clamp(min=1e-30)).tolist()
                    top = sorted(idx, key=lambda j: -rel[j])[:8]
                    first = (i, f'{len(idx)} of {len(rel)} gradient norms differ; largest relative differences: ' +
                             ', '.join(f'{grad_names[j]} {rel[j]:.2e}' for j in top) + '; smallest: ' +
                             ', '.join(f'{grad_names[j]} {rel[j]:.2e}' for j in sorted(idx, key=lambda j: rel[j])[:3]))
                else:
                    first = (i, na, va.tolist(), vb.tolist())
                break
        if first:
            break
    if first:
        bad += 1
        print('twin', rep, 'first difference:', first)
print(bad, 'of', TWINS, 'twins differ')
