"""Do small launches of the dense convolution kernel overlap across streams? N launches on one stream against the same N
spread over four streams (256 -> 128 channels on a 12 x 24 x 78 map: 108 tiles)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import dense_conv

dev = 'cuda:0'
torch.manual_seed(0)
xs = [torch.randn(12, 256, 24, 78, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]
w = torch.randn(128, 256, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
xa = [dense_conv.amax_bits(x) for x in xs]
wa = dense_conv.amax_bits(w)
ys = [torch.empty(12, 128, 24, 78, device=dev).contiguous(memory_format=torch.channels_last) for _ in range(4)]
streams = [torch.cuda.Stream() for _ in range(4)]
N = 40


def seq():
    for i in range(N):
        dense_conv._run(xs[i % 4], w, False, False, xa[i % 4], wa, ys[i % 4])


def par():
    main = torch.cuda.current_stream()
    for s in streams: s.wait_stream(main)
    for i in range(N):
        with torch.cuda.stream(streams[i % 4]):
            dense_conv._run(xs[i % 4], w, False, False, xa[i % 4], wa, ys[i % 4])
    for s in streams: main.wait_stream(s)


for name, fn in (('one stream', seq), ('four streams', par), ('one stream', seq), ('four streams', par)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn(); torch.cuda.synchronize()
    print(f'{name}: {(time.perf_counter() - t0) / N * 1e6:.0f} us per launch (pack + conv)', flush=True)
