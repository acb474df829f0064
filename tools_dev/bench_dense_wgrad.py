"""Time of the dense 3x3 weight-gradient kernel alone (gga_dense_wgrad3x3_planes through dense_conv._wgrad) at the
trunk's shapes, a 1 GiB fill between launches. Used with tools_dev/run_with_lib.py for kernel variants."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import dense_conv

dev = 'cuda:0'
trash = torch.empty(1 << 28, dtype=torch.float32, device=dev)


def timeit(fn, n=10):
    ts = []
    for i in range(n + 2):
        trash.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        if i >= 2: ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


for (B, C, Co, H, W) in ((16, 64, 64, 248, 216), (16, 128, 128, 124, 108), (16, 256, 256, 62, 54)):
    torch.manual_seed(0)
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(B, Co, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Co, C, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    xa, ga = dense_conv.amax_bits(x), dense_conv.amax_bits(gy)
    t = timeit(lambda: dense_conv._wgrad(x, gy, w, xa, ga))
    gf = 2.0 * B * H * W * C * Co * 9 / 1e9
    print(f'[{B},{C}->{Co},{H},{W}] wgrad {t * 1e3:.0f} us ({gf / t:.0f} TFLOP/s-eq)', flush=True)
