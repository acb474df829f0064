"""One 256 -> 256 tower convolution over the five FPN levels of the PGD head (bs 12): map by map against one launch."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import dense_conv

dev = 'cuda:0'
torch.manual_seed(0)
sizes = [(48, 156), (24, 78), (12, 39), (6, 20), (3, 10)]
conv = torch.nn.Conv2d(256, 256, 3, padding=1, bias=False).to(dev)
conv.weight.data = conv.weight.data.contiguous(memory_format=torch.channels_last)
xs = [torch.randn(12, 256, h, w, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for h, w in sizes]


def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


with torch.no_grad():
    xa = [dense_conv.amax_bits(x) for x in xs]
    wa = dense_conv.amax_bits(conv.weight)
    print(f'forward, map by map: {timeit(lambda: [dense_conv._run(x, conv.weight, False, False, a, wa) for x, a in zip(xs, xa)]):.0f} us')
    print(f'forward, one launch: {timeit(lambda: dense_conv._run_levels(xs, conv.weight, False, xa, wa)):.0f} us')


def fb(levels):
    ys = dense_conv.conv2d_levels(xs, conv) if levels else [dense_conv.conv2d(x, conv) for x in xs]
    torch.autograd.backward(ys, [y.detach() for y in ys])


print(f'forward + backward, map by map: {timeit(lambda: fb(False)):.0f} us')
print(f'forward + backward, one launch: {timeit(lambda: fb(True)):.0f} us')
