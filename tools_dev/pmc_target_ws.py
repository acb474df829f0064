"""PMC target: 12 stand-alone forward launches of the producer / consumer dense 3x3 kernel at the PointPillars shapes (no statistics)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch

from gga_amd import dense_conv

dense_conv.PLANES = 2
dev = 'cuda:0'
for B, cin, cout, H, W in [(16, 128, 128, 124, 108), (16, 64, 64, 248, 216)]:
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(12):
        dense_conv._run(x, w, False, False)
    torch.cuda.synchronize()
