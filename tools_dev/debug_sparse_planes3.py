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
orig_apply = sparse._conv_apply
captured = []
def conv_apply(x, rb, wp, n_rows, kvol, cin, cout, flip, y, x_amax=None, w_amax=None, stats=None, bn=None):
    if flip == 1 or (x.requires_grad is False and len(captured) < 100):
        captured.append(dict(x=x.detach().clone(), rb=rb, n_rows=n_rows, kvol=kvol, cin=cin, cout=cout, flip=flip, bn=bn, had_stats=stats is not None))
    return orig_apply(x, rb, wp, n_rows, kvol, cin, cout, flip, y, x_amax, w_amax, stats, bn)
dense_conv.PLANES = 3
e = copy.deepcopy(enc).to(DEV)
y = e(feats.to(DEV), coors.to(DEV), B)
captured.clear()
sparse._conv_apply = conv_apply
torch.manual_seed(1)
y.backward(torch.randn_like(y))
sparse._conv_apply = orig_apply
ws = [m.weight for m in e.modules() if isinstance(m, sparse.SparseConvolution)][::-1]
print(len(captured), 'backward-data calls')
for i, c in enumerate(captured):
    if (c['n_rows'], c['cin'], c['cout']) != (325, 64, 64):
        continue
    w = ws[i].detach().view(-1, ws[i].shape[-2], ws[i].shape[-1])       # [kvol, cin_fwd, cout_fwd]
    gy = c['x']
    rb = c['rb']
    # float64 reference of the backward-data product: gx[r] = sum_k gy[nbr[K-1-k][r]] @ W[K-1-k]^T ... (flip = 1)
    K = c['kvol']
    ref = torch.zeros(c['n_rows'], c['cout'], dtype=torch.float64, device=DEV)
    for k in range(K):
        kk = K - 1 - k
        idx = rb.nbr[kk].long()
        ok = idx >= 0
        ref[ok] += gy.double()[idx[ok]] @ w[k].double().t()
    outs = {}
    for planes in (2, 3):
        two = planes == 2
        g_amax = dense_conv._amax_bits(gy) if two else None
        w_amax = dense_conv._amax_bits(w.contiguous()) if two else None
        wt = sparse._pack_weight(w.contiguous(), K, c['cin'], c['cout'], 1, w_amax=w_amax)
        for use_bn in (False, True):
            if use_bn and c['bn'] is None:
                continue
            gx = torch.full((c['n_rows'], c['cout']), float('nan'), device=DEV)
            st = torch.empty((int(sparse._lib.lib().gga_sparse_conv_apply_tiles(c['n_rows'])), 2, c['cout']), dtype=torch.float64, device=DEV)
            orig_apply(gy, rb, wt, c['n_rows'], K, c['cin'], c['cout'], 1, gx, g_amax, w_amax, st, c['bn'] if use_bn else None)
            outs[(planes, use_bn)] = gx
    d = lambda u, v: float((u.double() - v.double()).norm() / v.double().norm())
    line = f'call {i}: bn given {c["bn"] is not None} |'
    for k_, v in outs.items():
        if not k_[1]:
            line += f' planes {k_[0]} plain vs f64 {d(v, ref):.2e};'
    if (2, True) in outs:
        line += f' fused 3 vs fused 2 {d(outs[(3, True)], outs[(2, True)]):.2e};'
        m2 = outs[(2, True)] != 0
        line += f' masked(plain2) vs fused2 {d(outs[(2, False)] * m2, outs[(2, True)]):.2e}; masked(plain3) vs fused3 {d(outs[(3, False)] * (outs[(3, True)] != 0), outs[(3, True)]):.2e}'
        diff = (outs[(3, True)] - outs[(2, True)]).abs()
        bad = diff > 1e-4 * outs[(2, True)].abs().max()
        line += f'; elements off {int(bad.sum())} in rows {sorted(set(bad.nonzero()[:, 0].tolist()))[:12]} cols {sorted(set(bad.nonzero()[:, 1].tolist()))[:12]}'
    print(line)
