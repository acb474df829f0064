"""Do an MFMA-bound kernel (dense weight gradient) and HBM-bound passes (BatchNorm backward) overlap when they are
launched on two streams? Sequential time against concurrent time at the trunk's 64-channel shape."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import dense_conv, functional as F

dev = 'cuda:0'
B, C, H, W = 16, 64, 248, 216
torch.manual_seed(0)
x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
gy = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn(C, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
xa, ga = dense_conv.amax_bits(x), dense_conv.amax_bits(gy)
bn = torch.nn.BatchNorm2d(C).to(dev)
y2 = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
g2 = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
side = torch.cuda.Stream()


def wgrad():
    dense_conv._wgrad(x, gy, w, xa, ga)


def bnpass():
    z = F.bn_act(y2, bn, relu=True)
    z.backward(g2)
    y2.grad = None


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def both():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        wgrad()
    bnpass()
    main.wait_stream(side)


tw, tb = timeit(wgrad), timeit(bnpass)
print(f'weight gradient {tw:.0f} us, BatchNorm forward + backward passes {tb:.0f} us, sum {tw + tb:.0f} us')
print(f'two streams: {timeit(both):.0f} us')
