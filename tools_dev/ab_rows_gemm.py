"""1x1 and kernel = stride transposed convolutions of the two LiDAR necks: the streaming kernel (gga_rows_gemm) against the
gather-GEMM with an arithmetic rule book, forward and backward (input + weight gradient) through the product's own modules,
alternating runs on the same box, two fp16 planes. A 2 GiB copy between timed calls keeps the caches cold as in the step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gga_amd import dense_conv, strided_conv
DEV = 'cuda:0'
dense_conv.PLANES = 2
junk = torch.empty(1 << 29, device=DEV)


def timed(fn, n=6):
    ts = []
    for i in range(n + 2):
        junk.add_(1.0)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2:
            ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


cases = [('pp   1x1  64->128 16x248x216', 'conv', 64, 128, 1, 16, 248, 216), ('pp   k2s2 128->128 16x124x108', 'deconv', 128, 128, 2, 16, 124, 108),
         ('pp   k4s4 256->128 16x62x54', 'deconv', 256, 128, 4, 16, 62, 54), ('sec  1x1  128->256 8x200x176', 'conv', 128, 256, 1, 8, 200, 176),
         ('sec  k2s2 256->256 8x100x88', 'deconv', 256, 256, 2, 8, 100, 88)]
for name, kind, cin, cout, k, B, H, W in cases:
    m = (torch.nn.Conv2d(cin, cout, 1, bias=False) if kind == 'conv' else torch.nn.ConvTranspose2d(cin, cout, k, k, bias=False)).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    res = {}
    for fast in (True, False, True, False):
        strided_conv.ROWS_GEMM = fast
        y = dense_conv.conv2d(x, m)
        g = torch.randn_like(y)
        fwd = timed(lambda: dense_conv.conv2d(x, m))

        def both():
            x.grad = None; m.weight.grad = None
            dense_conv.conv2d(x, m).backward(g)
        tot = timed(both)
        res.setdefault(fast, []).append((fwd, tot - fwd))
    f = lambda rows: ' / '.join(f'{a:6.1f} + {b:6.1f}' for a, b in rows)
    print(f'{name:32s} us fwd + bwd: rows_gemm {f(res[True])}   gather {f(res[False])}', flush=True)
