"""Stand-alone A/B of the consumer MFMA shapes of dense_conv_ws.hip: us per launch (median of 40 after 10, back to back) for the
shapes of the PointPillars step, on random and on all-zero activations (zero operands: the clock is not held down, the times
rank the instruction streams by cycles). Usage: ab_ws_shape.py"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch

from gga_amd import dense_conv

dense_conv.PLANES = 2
dev = 'cuda:0'
shapes = [(16, 128, 128, 124, 108), (16, 64, 64, 248, 216), (16, 256, 256, 62, 54), (16, 384, 64, 248, 216)]
for B, cin, cout, H, W in shapes:
    torch.manual_seed(0)
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    for data in ('random', 'zeros'):
        x = (torch.randn(B, cin, H, W, device=dev) if data == 'random' else torch.zeros(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        if data == 'zeros':
            x[0, 0, 0, 0] = 1.0
        out = {}
        for rep in range(2):
            for m in ('32', '16'):
                os.environ['GGA_DC_WS_MFMA'] = m
                for _ in range(10):
                    dense_conv._run(x, w, False, True)
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
                ev[0].record()
                for i in range(40):
                    dense_conv._run(x, w, False, True)
                    ev[i + 1].record()
                torch.cuda.synchronize()
                ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
                out.setdefault(m, []).append(round(ts[20] * 1e3, 1))
        print(f'{B}x{cin}->{cout}x{H}x{W} {data:6s} 32x32x16 {out["32"]} us   16x16x32 {out["16"]} us   ratio {out["16"][1] / out["32"][1]:.3f}')
