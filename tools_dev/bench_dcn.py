"""DCNv2 pieces at the PGD head's level-0 shape (12 x 256 x 96 x 312): im2col, the two GEMMs, col2im."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import _lib, functional as F
dev = 'cuda:0'
B, C, H, W = (int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (12, 256, 96, 312)))
L = _lib.lib()
x = torch.randn(B, H, W, C, device=dev)
off = torch.randn(B, 18, H, W, device=dev) * float(sys.argv[5] if len(sys.argv) > 5 else 0.5)
mask = torch.rand(B, 9, H, W, device=dev)
col = torch.empty(B * H * W, 9 * C, device=dev)
gcol = torch.randn_like(col)
gx, goff, gm = torch.empty_like(x), torch.empty_like(off), torch.empty_like(mask)
w = torch.randn(256, 9 * C, device=dev) * 0.02
geom = (B, H, W, C, 3, 3, 1, 1, 1, 1, 1, 1)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print('im2col  %.3f ms' % t(lambda: L.gga_dcn_im2col(F._p(x), F._p(off), F._p(mask), *geom, F._p(col), F._stream())))
print('gemm    %.3f ms' % t(lambda: col @ w.t()))
print('col2im  %.3f ms' % t(lambda: L.gga_dcn_col2im(F._p(x), F._p(off), F._p(mask), F._p(gcol), *geom, F._p(gx), F._p(goff), F._p(gm), F._stream())))
print('col2im (no grad_x) %.3f ms' % t(lambda: L.gga_dcn_col2im(F._p(x), F._p(off), F._p(mask), F._p(gcol), *geom, None, F._p(goff), F._p(gm), F._stream())))
print('bytes: col %.0f MB, x %.0f MB' % (col.numel() * 4 / 1e6, x.numel() * 4 / 1e6))
