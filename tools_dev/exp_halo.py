"""Experiment: the halo form of the SubM gather-GEMM (sp_conv_halo_kernel) against sp_conv_x9_kernel on the levels of the
shipped config's bench batch (bs 8): same product, time, agreement, size of the halos."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import Config, build_model, synthetic, _lib, dense_conv
from gga_amd import functional as F
from gga_amd.sparse import SparseConvTensor, _Halo, _pack_weight
DEV = 'cuda:0'
BS = 8


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

for planes in (2,):
    dense_conv.PLANES = planes
    for li, (lvl, C_) in enumerate(levels):
        if C_ < 64:
            continue
        nrow = lvl.n
        feats = torch.randn(nrow, C_, device=DEV)
        w = torch.randn(27, C_, C_, device=DEV) * 0.05
        rb = lvl.subm_rulebook((3, 3, 3))
        torch.cuda.synchronize()
        t0 = time.time()
        halo = _Halo(lvl.coors, rb)
        torch.cuda.synchronize()
        t_build = (time.time() - t0) * 1e3
        cnt = halo.counts.float()
        print(f'=== planes {planes} level {li + 1}: {nrow} rows, C {C_}, grid {lvl.shape}; halo rows per 256-row tile: mean {cnt.mean():.0f} '
              f'max {int(cnt.max())}, tiles over 512: {float((cnt > 512).float().mean()):.3f}; tables built in {t_build:.2f} ms')
        two = planes == 2
        x_amax = dense_conv._amax_bits(feats) if two else None
        w_amax = dense_conv._amax_bits(w) if two else None
        for flip in (0, 1):
            wp = _pack_weight(w, 27, C_, C_, flip, w_amax=w_amax)
            y0 = torch.empty(nrow, C_, device=DEV)
            y1 = torch.zeros(nrow, C_, device=DEV)
            st0 = torch.empty((int(L.gga_sparse_conv_apply_tiles(nrow)), 2, C_), dtype=torch.float64, device=DEV)
            st1 = torch.empty_like(st0)
            f0 = lambda: L.gga_sparse_conv_apply_stats(F._p(feats), F._p(rb.nbr), F._p(wp), F._p(rb.perm), F._p(rb.mask), nrow, 27, C_, C_, flip,
                                                       F._p(y0), C_, planes, F._p(x_amax), F._p(w_amax), F._p(st0), F._stream())
            f1 = lambda: L.gga_sparse_conv_apply_halo(F._p(feats), F._p(wp), F._p(halo.tile_rows), F._p(halo.counts), halo.capacity, F._p(halo.halo_rows),
                                                      F._p(halo.local_map), nrow, halo.n_tiles, 27, C_, C_, flip, F._p(y1), C_, planes,
                                                      F._p(x_amax), F._p(w_amax), F._p(st1), None, 0, None, None, None, None, F._stream())
            assert f0() == 0
            rc = f1()
            _lib.check(rc, "halo")
            torch.cuda.synchronize()
            err = float((y0 - y1).abs().max() / y0.abs().max())
            serr = float((st0.sum(0) - st1.sum(0)).abs().max() / st0.sum(0).abs().max())
            print(f'  flip {flip}: x9 {timeit(f0):7.1f} us   halo {timeit(f1):7.1f} us   max |dy| / max |y| {err:.2e}  stats {serr:.2e}')
        if os.environ.get('GGA_HALO_TIMES'):
            import ctypes
            raw = ctypes.CDLL(_lib.LIB_PATH)
            buf = (ctypes.c_ulonglong * 8)()
            raw.gga_debug_halo_times(buf, 1)
            f1(); torch.cuda.synchronize()
            raw.gga_debug_halo_times(buf, 1)
            names = ['prologue', 'wait weights', 'barrier', 'chunk switch', 'stage body', 'epilogue']
            tot = sum(buf[:6])
            print('   wave-0 phase shares (wall_clock64 ticks, summed over tiles): ' + ', '.join(f'{n} {buf[i] / tot:.3f}' for i, n in enumerate(names)),
                  f'| ticks per tile {tot / halo.n_tiles:.0f}')
