#!/usr/bin/env python
"""Why is the scatter canvas kernel slower inside a train step than back to back? Times the
kernel (library timing hooks) after different memory histories of the canvas block."""
import ctypes as C
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
from gga_amd import _lib
from gga_amd import functional as F
dev = torch.device('cuda:0')
B, Cc, ny, nx, M = 16, 64, 496, 432, 16000
g = torch.Generator().manual_seed(0)
coors = []
for b in range(B):
    cells = torch.randperm(ny * nx, generator=g)[:M]
    coors.append(torch.stack([torch.full((M,), b), torch.zeros(M, dtype=torch.long), cells // nx, cells % nx], 1))
coors = torch.cat(coors).int().to(dev)
feats = torch.randn(B * M, Cc, device=dev)
L = _lib.lib()
big = torch.empty(1 << 29, device=dev)      # 2 GiB
big2 = torch.empty(1 << 29, device=dev)
conv = torch.nn.Conv2d(64, 64, 3, padding=1).to(dev).to(memory_format=torch.channels_last)


def run(name, between, n=8):
    _lib.check(L.gga_pillar_scatter_timing_begin(n), 'b')
    for _ in range(n):
        y = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=True)
        between(y)
        del y
    torch.cuda.synchronize()
    buf = (C.c_float * 256)()
    k = L.gga_pillar_scatter_timing_collect(buf, 256)
    v = [buf[i] * 1e3 for i in range(k)]
    print(f'{name:58s} ' + ' '.join(f'{x:6.1f}' for x in v))


run('back to back', lambda y: None)
run('canvas.sum() between', lambda y: y.sum())
run('2 GiB copy between (flush MALL)', lambda y: big2.copy_(big))
run('canvas.sum() + 2 GiB copy', lambda y: (y.sum(), big2.copy_(big)))
run('conv fwd reading the canvas', lambda y: conv(y))
run('conv fwd + 2 GiB copy', lambda y: (conv(y), big2.copy_(big)))
run('canvas.mul_(1) (normal-store rewrite) between', lambda y: y.mul_(1.0))
run('canvas.mul_(1) + 2 GiB copy', lambda y: (y.mul_(1.0), big2.copy_(big)))
