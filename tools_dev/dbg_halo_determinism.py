"""Is the halo kernel deterministic? The same launch (and the same tiling build) repeated on the bench batch's 128-channel level:
bitwise comparison of outputs, statistics and tables."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, _lib, dense_conv
from gga_amd import functional as F
from gga_amd.sparse import SparseConvTensor, _Halo, _pack_weight
DEV = 'cuda:0'
BS = 8
dense_conv.PLANES = 2
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
cfg = Config.fromfile(os.path.join(root, 'configs/gga/gga_kitti_config.py'))
torch.manual_seed(0)
model = build_model(cfg.model).to(DEV).train()
b = synthetic.make_batch(BS, n_points=20000, pc_range=synthetic.RANGE_SECOND)
pts = [p.to(DEV) for p in b['points']]
v, n, c = model.voxelize(pts)
f = model.pts_voxel_encoder(v, n, c)
enc = model.pts_middle_encoder
L = _lib.lib()
with torch.no_grad():
    x = SparseConvTensor(f, c.int(), enc.sparse_shape, BS)
    x = enc.conv_input(x)
    for layer in enc.encoder_layers:
        for m in layer:
            x = m(x)
lvl = x._level
nrow, C_ = lvl.n, 128
feats = torch.randn(nrow, C_, device=DEV)
w = torch.randn(27, C_, C_, device=DEV) * 0.05
rb = lvl.subm_rulebook((3, 3, 3))
h0 = _Halo(lvl.coors, rb)
h1 = _Halo(lvl.coors, rb)
print('tables equal:', torch.equal(h0.tile_rows, h1.tile_rows), torch.equal(h0.counts, h1.counts), torch.equal(h0.local_map, h1.local_map),
      torch.equal(h0.halo_rows[:, :300], h1.halo_rows[:, :300]))
x_amax, w_amax = dense_conv._amax_bits(feats), dense_conv._amax_bits(w)
# a second stream keeps small kernels running beside the launches (what the prefetch of the next batch does in the step)
side = torch.cuda.Stream(priority=-1)
noise = torch.randn(1 << 20, device=DEV)
keys = torch.randint(0, 1 << 30, (1 << 18,), device=DEV)


def disturb(n=40):
    # the real thing: the point-only front of a batch (voxelizer, sparse index plan) on the side stream
    with torch.cuda.stream(side):
        model.prepare_inputs(pts)


for rep in range(12):
    hb = _Halo(lvl.coors, rb)
    disturb(10)
    torch.cuda.synchronize()
    ok = torch.equal(h0.tile_rows, hb.tile_rows) and torch.equal(h0.counts, hb.counts) and torch.equal(h0.local_map, hb.local_map)
    cnt = h0.counts.long()
    valid = torch.arange(h0.capacity, device=DEV)[None, :] < cnt[:, None]
    ok = ok and torch.equal(h0.halo_rows[valid], hb.halo_rows[valid])
    if not ok:
        print('  tiling differs under concurrency, repeat', rep)
print('tiling repeated under concurrency')
for flip in (0, 1):
    wp = _pack_weight(w, 27, C_, C_, flip, w_amax=w_amax)
    ref = None
    bad = 0
    for it in range(40):
        ys = [torch.full((nrow, C_), float('nan'), device=DEV) for _ in range(4)]
        st = torch.empty((int(L.gga_sparse_conv_apply_tiles(nrow)), 2, C_), dtype=torch.float64, device=DEV)
        for y in ys:                       # four launches queued, then the front beside them
            _lib.check(L.gga_sparse_conv_apply_halo(F._p(feats), F._p(wp), F._p(h0.tile_rows), F._p(h0.counts), h0.capacity, F._p(h0.halo_rows),
                                                    F._p(h0.local_map), nrow, h0.n_tiles, 27, C_, C_, flip, F._p(y), C_, 2, F._p(x_amax),
                                                    F._p(w_amax), F._p(st), None, 0, None, None, None, None, F._stream()), 'halo')
        if it > 5:
            disturb()
        torch.cuda.synchronize()
        for y in ys[:-1]:
            if not torch.equal(y, ys[-1]):
                print('   launches of one burst differ')
        y = ys[-1]
        if ref is None:
            ref = (y.clone(), st.clone())
        else:
            same = torch.equal(y, ref[0]) and torch.equal(st, ref[1])
            if not same:
                bad += 1
                d = (y - ref[0]).abs()
                print(f'  flip {flip} run {it}: differs, max |d| {float(d.max()):.3e} at {int((d > 0).sum())} elements, max |y| {float(ref[0].abs().max()):.3e}')
    print(f'flip {flip}: {bad} of 39 repeats differ')
