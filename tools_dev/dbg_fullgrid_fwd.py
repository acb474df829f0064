"""Forward of every sparse convolution of the full-grid encoder test, in situ, against float64 from the same operands."""
import os, sys
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, 'tests')]
from gga_amd import Config, synthetic, dense_conv, sparse, functional as F
from gga_amd.registry import build_middle_encoder
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
real = sparse.SparseConvolution._run


def run(self, feats, w, rb, rb_t, n_out):
    y = real(self, feats, w, rb, rb_t, n_out)
    st = getattr(y, 'bn_partials', None)
    calls.append((self, feats.detach(), w.detach(), rb.nbr if hasattr(rb, 'nbr') else rb, y.detach(), None if st is None else st.detach().clone()))
    return y
sparse.SparseConvolution._run = run
with torch.no_grad():
    enc(feats, c, B)
names = {m: n_ for n_, m in enc.named_modules()}
for m, f, w, nbr, y, st in calls:
    nbr = nbr.long()
    x64, w64 = f.double(), w.double()
    yy = x64.new_zeros(nbr.shape[1], w64.shape[-1])
    for k in range(nbr.shape[0]):
        rows = (nbr[k] >= 0).nonzero()[:, 0]
        if len(rows):
            yy = yy.index_add(0, rows, x64[nbr[k, rows]] @ w64[k])
    d = (y.double() - yy)
    rel = float(d.norm() / yy.norm())
    rowerr = d.norm(dim=1) / yy.norm(dim=1).clamp_min(1e-30)
    print(f'{names[m]:42s} rows {nbr.shape[1]:7d} cin {w.shape[-2]:3d} cout {w.shape[-1]:3d}  rel err {rel:.2e}  max abs err {float(d.abs().max()):.2e} (|y| max {float(yy.abs().max()):.2e})  '
          + (f'stats: sum rel err {float(((st[:, 0].sum(0)) - y.double().sum(0)).abs().max() / y.double().sum(0).abs().max()):.1e}, '
             f'sum of squares rel err {float(((st[:, 1].sum(0)) - (y.double() ** 2).sum(0)).abs().max() / (y.double() ** 2).sum(0).abs().max()):.1e}; ' if st is not None else 'no stats; ') +
          f'rows with rel err > 1e-5: {int((rowerr > 1e-5).sum())}, > 1e-3: {int((rowerr > 1e-3).sum())}; worst row {float(rowerr.max()):.2e}')
