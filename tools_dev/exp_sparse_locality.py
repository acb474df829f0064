"""Experiment: what do spatially ordered rows buy the gather-GEMM kernels at the 128-channel level of the shipped config?
Variants of the SAME SubM 128->128 convolution (bench batch, bs 8):
  base     rows in the level's own order, rows processed in global mask order (what the product does)
  sorted   rows of the level re-ordered spatially (b, y/T, x/T, z, y, x), global mask order
  region   + rows processed region by region (R consecutive rows) in mask order inside a region
Forward (sp_conv_x9_kernel) under GGA_SP_TILE_ORDER as set in the environment, and the weight gradient."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, _lib, dense_conv
from gga_amd import functional as F
from gga_amd.sparse import SparseConvTensor, _Level, _pack_weight
DEV = 'cuda:0'
BS = 8
dense_conv.PLANES = 2


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


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
    levels = []
    for layer in enc.encoder_layers:
        for m in layer:
            x = m(x)
        levels.append((x._level, x.features.shape[1]))


def unions(mask_sorted, g, kvol=27):
    n = (len(mask_sorted) // g) * g
    t = mask_sorted[:n].view(-1, g).long()
    u = torch.zeros(t.shape[0], dtype=torch.long, device=t.device)
    for i in range(g):
        u |= t[:, i]
    return float(sum(((u >> k) & 1) for k in range(kvol)).float().mean())


for li, (lvl, C_) in enumerate(levels):
    if li != int(os.environ.get("GGA_EXP_LEVEL", "3")):
        continue
    coors = lvl.coors
    nrow = lvl.n
    feats = torch.randn(nrow, C_, device=DEV)
    w = torch.randn(27, C_, C_, device=DEV) * 0.05
    print(f'=== level {li + 1}: {nrow} rows, C {C_}, grid {lvl.shape}')
    for T in ((None, 8) if os.environ.get('GGA_EXP_FEW') else (None, 8, 4)):
        if T is None:
            order = torch.arange(nrow, device=DEV)
            name = 'base  '
        else:
            cc = coors.long()
            D, H, W = lvl.shape
            key = ((((cc[:, 0] * (H // T + 1) + cc[:, 2] // T) * (W // T + 1) + cc[:, 3] // T) * D + cc[:, 1]) * H + cc[:, 2]) * W + cc[:, 3]
            order = torch.argsort(key)
            name = f'sort{T:2d}'
        lv = _Level(coors[order].contiguous(), lvl.shape, BS)
        rb = lv.subm_rulebook((3, 3, 3))
        xs = feats[order].contiguous()
        x_amax, w_amax = dense_conv._amax_bits(xs), dense_conv._amax_bits(w)
        wp = _pack_weight(w, 27, C_, C_, 0, w_amax=w_amax)
        y = torch.empty(nrow, C_, device=DEV)
        pairs = float((rb.nbr >= 0).sum()) / nrow
        for R in ((0,) if T is None else (-1,)) if os.environ.get('GGA_EXP_FEW') else (0, -1):
            if R == 0:
                perm = rb.perm
            elif R == -1:
                perm = torch.arange(nrow, device=DEV, dtype=torch.int32)          # rows as they lie (spatial order when sorted)
            else:
                if T is None:
                    continue
                k2 = (torch.arange(nrow, device=DEV) // R) * (1 << 32) + (rb.mask.long() & 0xFFFFFFFF)
                perm = torch.argsort(k2, stable=True).int()
            ms = rb.mask[perm.long()] & 0x7FFFFFF
            t_f = timeit(lambda: L.gga_sparse_conv_apply_planes(F._p(xs), F._p(rb.nbr), F._p(wp), F._p(perm), F._p(rb.mask), nrow, 27, C_, C_, 0,
                                                               F._p(y), C_, 2, F._p(x_amax), F._p(w_amax), F._stream()))
            print(f'  {name} region {R:6d}: pairs/row {pairs:.2f} offsets per 128-row tile {unions(ms, 128):5.2f} per 256 {unions(ms, 256):5.2f} | forward {t_f:7.1f} us')
        gw = torch.empty_like(w)
        g = torch.randn(nrow, C_, device=DEV)
        g_amax = dense_conv._amax_bits(g)
        ws = torch.empty(L.gga_sparse_conv_wgrad_workspace_bytes(nrow, 27, C_, C_), dtype=torch.uint8, device=DEV)
        t_w = timeit(lambda: L.gga_sparse_conv_wgrad_planes(F._p(xs), C_, F._p(g), C_, F._p(rb.nbr), nrow, 27, C_, C_, F._p(gw), 2, F._p(x_amax), F._p(g_amax),
                                                           F._p(ws), ws.numel(), F._stream()))
        print(f'  {name} weight gradient {t_w:7.1f} us')
