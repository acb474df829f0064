#!/usr/bin/env python
"""3x3 stride-1 conv on channels-last activations: the bf16x9 gather-GEMM path vs MIOpen (fp32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dense_conv_x9 as D
dev = 'cuda:0'
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for (B, C, H, W) in ((16, 64, 248, 216), (16, 128, 124, 108)):
    conv = torch.nn.Conv2d(C, C, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    g = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    y0 = conv(x); y1 = D.conv3x3(x, conv.weight)
    err = float((y0 - y1).abs().max() / y0.abs().max())
    gx0, gw0 = torch.autograd.grad(y0, (x, conv.weight), g); gx1, gw1 = torch.autograd.grad(y1, (x, conv.weight), g)
    e2 = float((gx0 - gx1).abs().max() / gx0.abs().max()); e3 = float((gw0 - gw1).abs().max() / gw0.abs().max())
    fl = 2.0 * B * H * W * C * C * 9
    with torch.no_grad():
        tm = timeit(lambda: conv(x)); tx = timeit(lambda: D.conv3x3(x, conv.weight))
    def fb(fn):
        y = fn(); gx, gw = torch.autograd.grad(y, (x, conv.weight), g)
    tmb = timeit(lambda: fb(lambda: conv(x))); txb = timeit(lambda: fb(lambda: D.conv3x3(x, conv.weight)))
    print(f'C={C} {H}x{W}: fwd MIOpen {tm:.3f} ms ({fl/tm/1e9:.0f} TF/s), bf16x9 {tx:.3f} ms ({fl/tx/1e9:.0f} TF/s) | fwd+bwd MIOpen {tmb:.2f} ms, bf16x9 {txb:.2f} ms | '
          f'max rel diff y {err:.1e} gx {e2:.1e} gw {e3:.1e}')
