"""Where do the gradients of the small-grid SparseEncoder first differ between the two plane forms? Gradient w.r.t. the
features after every module (conv / BasicBlock / SparseSequential), compared 3 planes vs 2 planes, in backward order."""
import copy, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import torch
from gga_amd import dense_conv, sparse
from gga_amd.sparse_encoder import SparseEncoder
from test_sparse_gpu import _coords
DEV = 'cuda:0'
torch.manual_seed(0)
shape, B = (41, 40, 32), 2
enc = SparseEncoder(in_channels=4, sparse_shape=list(shape), output_channels=128, order=('conv', 'norm', 'act'),
                    encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                    encoder_paddings=((0, 0, 1), (0, 0, 1), (0, 0, [0, 1, 1]), (0, 0)), block_type='basicblock')
enc.train()
coors = _coords(B, shape, 900, seed=5)
feats = torch.randn(len(coors), 4)
g = None
res = {}
for planes in (2, 3):
    dense_conv.PLANES = planes
    e = copy.deepcopy(enc).to(DEV)
    grads, fw = {}, {}
    hooks = []
    for name, m in e.named_modules():
        if isinstance(m, sparse.SparseModule) and name:
            def hook(mod, inp, out, name=name):
                if isinstance(out, sparse.SparseConvTensor) and out.features.requires_grad:
                    fw[name] = out.features.detach().clone()
                    out.features.register_hook(lambda gr, name=name: grads.__setitem__(name, gr.detach().clone()))
            hooks.append(m.register_forward_hook(hook))
    y = e(feats.to(DEV), coors.to(DEV), B)
    if g is None:
        g = torch.randn_like(y)
    y.backward(g)
    res[planes] = (fw, grads, {n: p.grad.clone() for n, p in e.named_parameters()})
f2, g2, p2 = res[2]
f3, g3, p3 = res[3]
for n in f2:
    a, b = g2.get(n), g3.get(n)
    ge = float((a - b).norm() / a.norm()) if a is not None and b is not None else float('nan')
    print(f'{n:45s} rows {f2[n].shape[0]:5d} C {f2[n].shape[1]:3d} fwd diff {float((f2[n] - f3[n]).norm() / f2[n].norm()):.2e}  grad diff {ge:.2e}')

# ---- finer: every backward-data result, cloned right after the launch and again after the weight gradient of the same call
print('per-call backward-data results (3 planes vs 2 planes)')
calls = {}
orig_apply, orig_wgrad = sparse._conv_apply, sparse.conv_wgrad
for planes in (2, 3):
    dense_conv.PLANES = planes
    log = calls[planes] = []
    state = {}
    def conv_apply(x, rb, wp, n_rows, kvol, cin, cout, flip, y, *a, **k):
        r = orig_apply(x, rb, wp, n_rows, kvol, cin, cout, flip, y, *a, **k)
        state['last'] = (y, y.detach().clone(), (n_rows, kvol, cin, cout, flip))
        return r
    def conv_wgrad(x, gy, nbr, n_rows, kvol, cin, cout, gw, *a, **k):
        r = orig_wgrad(x, gy, nbr, n_rows, kvol, cin, cout, gw, *a, **k)
        torch.cuda.synchronize()
        y, before, sig = state['last']
        log.append((sig, before, y.detach().clone(), gw.detach().clone()))
        return r
    sparse._conv_apply, sparse.conv_wgrad = conv_apply, conv_wgrad
    e = copy.deepcopy(enc).to(DEV)
    y = e(feats.to(DEV), coors.to(DEV), B)
    log.clear()
    y.backward(g)
sparse._conv_apply, sparse.conv_wgrad = orig_apply, orig_wgrad
for (s2, b2, a2, w2), (s3, b3, a3, w3) in zip(calls[2], calls[3]):
    d = lambda u, v: float((u - v).norm() / (v.norm() + 1e-30))
    print(f'{str(s2):32s} gx right after launch: 3 vs 2 planes {d(b3, b2):.2e}; after wgrad vs before (2 planes) {d(a2, b2):.2e} (3 planes) {d(a3, b3):.2e}; gw 3 vs 2 {d(w3, w2):.2e}')

print('mask check of the layer3.0.conv1 backward-data result against the forward ReLU of layer2.2')
for planes in (2, 3):
    z = res[planes][0]['encoder_layers.encoder_layer2.2']
    sig, before, after, gw = [c for c in calls[planes] if c[0] == (325, 27, 64, 64, 1)][3]
    live = z > 0
    print(f'planes {planes}: z>0 {int(live.sum())} of {z.numel()}; gx != 0 where z <= 0: {int(((before != 0) & ~live).sum())}; gx == 0 where z > 0: {int(((before == 0) & live).sum())}')
b2 = [c for c in calls[2] if c[0] == (325, 27, 64, 64, 1)][3][1]
b3 = [c for c in calls[3] if c[0] == (325, 27, 64, 64, 1)][3][1]
diff = (b3 - b2).abs()
bad = diff > 1e-4 * b2.abs().max()
print('elements off', int(bad.sum()), 'rows', sorted(set(bad.nonzero()[:, 0].tolist()))[:20], 'cols', sorted(set(bad.nonzero()[:, 1].tolist()))[:70])
print('values', b2[bad][:8].tolist(), b3[bad][:8].tolist())
