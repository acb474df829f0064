"""Every sparse convolution's backward of the full-grid encoder test, in situ: the kernels' (gx, gw) against a float64
pair-list recomputation from the SAME operands (feats, w, gy) the call received."""
import copy, os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
from gga_amd import Config, synthetic, dense_conv, sparse, functional as F
from gga_amd.registry import build_middle_encoder
from oracle import sparse_ref as SR
dense_conv.PLANES = int(sys.argv[1]) if len(sys.argv) > 1 else 2
DEV = 'cuda:0'
cfg = Config.fromfile(os.path.join(REPO, 'configs', 'gga', 'gga_kitti_config.py'))
torch.manual_seed(0)
enc = build_middle_encoder(cfg.model.pts_middle_encoder).train().to(DEV)
B = 4
batch = synthetic.make_batch(B, n_points=20000, pc_range=synthetic.RANGE_SECOND)
vl = cfg.model.pts_voxel_layer
v, n, c, _ = F.hard_voxelize_batch([p.to(DEV) for p in batch['points']], vl.voxel_size, vl.point_cloud_range, vl.max_num_points, vl.max_voxels[0])
feats = F.voxel_mean(v, n, 4)
calls = []
real_bwd = sparse._SparseConvFn.backward


def bwd(ctx, gy, _gstats=None):
    out = real_bwd(ctx, gy, _gstats)
    f, w = ctx.saved_tensors
    calls.append(dict(f=f.detach(), w=w.detach(), gy=gy.detach().clone(), gx=None if out[0] is None else out[0].detach().clone(),
                      gw=None if out[1] is None else out[1].detach().clone(), nbr=ctx.rb.nbr, amax=ctx.amax))
    return out
sparse._SparseConvFn.backward = staticmethod(bwd)
y = enc(feats, c, B)
torch.manual_seed(1)
g = torch.randn_like(y)
y.backward(g)
torch.cuda.synchronize()
names = [n_ for n_, m in enc.named_modules() if isinstance(m, sparse.SparseConvolution)][::-1]
for name, r in zip(names, calls):
    nbr = r['nbr'].long()
    x64, w64, g64 = r['f'].double().requires_grad_(True), r['w'].double().requires_grad_(True), r['gy'].double()
    yy = x64.new_zeros(nbr.shape[1], w64.shape[-1])
    for k in range(nbr.shape[0]):
        rows = (nbr[k] >= 0).nonzero()[:, 0]
        if len(rows):
            yy = yy.index_add(0, rows, x64[nbr[k, rows]] @ w64[k])
    yy.backward(g64)
    e = lambda a, b: float((a.double() - b).norm() / b.norm().clamp_min(1e-300))
    if r['gx'] is not None:
        plain = e(r['gx'], x64.grad)
        if plain > 1e-3:          # the launch carried the BatchNorm-backward epilogue: gx is masked by the ReLU of the layer below = (f > 0)
            m = r['f'] > 0
            masked = x64.grad * m
            mism = int(((r['gx'] != 0) ^ m).sum())
            print(f"   masked: gx rel err vs float64 * (f > 0): {e(r['gx'], masked):.2e}; elements where (gx != 0) xor (f > 0): {mism} of {m.numel()}; "
                  f"their share of |gx|^2: {float((r['gx'].double()[(r['gx'] != 0) ^ m] ** 2).sum() / (r['gx'].double() ** 2).sum()):.2e}")
    ax = [None if a is None else float(torch.tensor(int(a.cpu()[0]), dtype=torch.int32).view(torch.float32)) for a in r['amax']]
    print(f"{name:42s} rows {nbr.shape[1]:7d} cin {r['w'].shape[-2]:3d} cout {r['w'].shape[-1]:3d}  gx rel err {e(r['gx'], x64.grad) if r['gx'] is not None else -1:.2e}  "
          f"gw rel err {e(r['gw'], w64.grad):.2e}   |gy| max {float(r['gy'].abs().max()):.3e} amax x / w {ax}  |f| max {float(r['f'].abs().max()):.3e}")
