#!/usr/bin/env python
"""Would one 64->960 conv beat the 15 separate 64->64 first-layer convs of the task heads (MIOpen, fp32, NHWC)?"""
import torch, time
dev = 'cuda:0'
B, H, W = 16, 248, 216
x = torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
convs = [torch.nn.Conv2d(64, 64, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last) for _ in range(15)]
big = torch.nn.Conv2d(64, 960, 3, padding=1, bias=False).to(dev).to(memory_format=torch.channels_last)
gs = [torch.randn(B, 64, H, W, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(15)]
gbig = torch.randn(B, 960, H, W, device=dev).contiguous(memory_format=torch.channels_last)
def sep_fwd():
    with torch.no_grad(): return [c(x) for c in convs]
def big_fwd():
    with torch.no_grad(): return big(x)
def sep_fb():
    ys = [c(x) for c in convs]
    torch.autograd.backward(ys, gs)
    x.grad = None
def big_fb():
    y = big(x); y.backward(gbig); x.grad = None
print(f'fwd: 15 separate {timeit(sep_fwd):.2f} ms, one 64->960 {timeit(big_fwd):.2f} ms')
print(f'fwd+bwd: 15 separate {timeit(sep_fb):.2f} ms, one 64->960 {timeit(big_fb):.2f} ms')
