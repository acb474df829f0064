"""Backward of a head-branch tail (BatchNorm -> ReLU -> 3x3 conv to 1..3 channels) at the bench size:
gga_head_tail_bwd (input gradient rebuilt in registers) against backward-data (MIOpen) + gga_bn_relu_bwd."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gga_amd import _lib
from gga_amd import functional as F
DEV = 'cuda:0'
L = _lib.lib()


def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, H, W, C = 16, 248, 216, 64
rows = B * H * W
x = torch.randn(B, C, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
gamma, beta = torch.rand(C, device=DEV) + 0.5, torch.rand(C, device=DEV) - 0.5
rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
saved, ss = torch.empty(2 * C, device=DEV), torch.empty(2 * C, device=DEV)
ws = torch.empty(L.gga_bn_relu_workspace_bytes(rows, C), dtype=torch.uint8, device=DEV)
_lib.check(L.gga_bn_stats(F._p(x), F._p(gamma), F._p(beta), F._p(rm), F._p(rv), rows, C, 1e-3, 0.01, 1, F._p(saved), F._p(ss), F._p(ws),
                          ws.numel(), F._stream()), 'stats')
for cout in (1, 2, 3):
    w = torch.randn(cout, C, 3, 3, device=DEV) * 0.05
    gy = torch.randn(B, cout, H, W, device=DEV)
    gx, gg, gb = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    gx2, gg2, gb2 = torch.empty_like(x), torch.empty(C, device=DEV), torch.empty(C, device=DEV)

    def fused():
        _lib.check(L.gga_head_tail_bwd(F._p(gy), F._p(x), C, F._p(ss), F._p(gamma), F._p(saved), F._p(w), B, H, W, C, cout, F._p(gx), C,
                                       F._p(gg), F._p(gb), None, F._p(ws), ws.numel(), F._stream()), 'tail')

    def split():
        gh = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
        gh = gh.contiguous(memory_format=torch.channels_last)
        _lib.check(L.gga_bn_relu_bwd(F._p(gh), F._p(x), F._p(ss), F._p(gamma), F._p(saved), rows, C, 2, F._p(gx2), None, F._p(gg2), F._p(gb2),
                                     F._p(ws), ws.numel(), F._stream()), 'bn bwd')
    fused(); split()
    torch.cuda.synchronize()
    print(f'cout {cout}: fused {timeit(fused):6.0f} us | backward-data + BatchNorm backward {timeit(split):6.0f} us | '
          f'max |dx diff| {float((gx - gx2).abs().max()):.2e} of {float(gx2.abs().max()):.2e}, dgamma rel {float((gg - gg2).abs().max() / gg2.abs().max()):.1e}')
