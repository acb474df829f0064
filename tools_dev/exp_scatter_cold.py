#!/usr/bin/env python
"""Scatter canvas kernel with a cold memory system (2 GiB of unrelated traffic between launches),
next to a plain fill of the same canvas. Env: GGA_SCATTER_NHWC_UNROLL, GGA_SCATTER_NHWC_NT."""
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
big = torch.empty(1 << 29, device=dev); big2 = torch.empty(1 << 29, device=dev)
algo = B * M * Cc * 4 + B * M * 16 + B * Cc * ny * nx * 4
n = 8
for cl in (True, False):
    _lib.check(L.gga_pillar_scatter_timing_begin(n), 'b')
    for _ in range(n):
        y = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=cl)
        big2.copy_(big); del y
    torch.cuda.synchronize()
    buf = (C.c_float * 256)()
    k = L.gga_pillar_scatter_timing_collect(buf, 256)
    v = sorted(buf[i] * 1e3 for i in range(k))
    print(f"scatter {'nhwc' if cl else 'nchw'} U={os.environ.get('GGA_SCATTER_NHWC_UNROLL','1')} NT={os.environ.get('GGA_SCATTER_NHWC_NT','1')} V={os.environ.get('GGA_SCATTER_VARIANT','-')}: median {v[len(v)//2]:.1f} us = {algo / v[len(v)//2] / 1e3:.0f} GB/s")
if os.environ.get('FILL'):
    canvas = torch.empty(B * Cc * ny * nx, device=dev)
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); canvas.zero_(); e1.record(); big2.copy_(big); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f'torch zero_ of the canvas (cold): median {ts[len(ts)//2]:.1f} us = {canvas.numel() * 4 / ts[len(ts)//2] / 1e3:.0f} GB/s')
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); canvas.zero_(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f'torch zero_ of the canvas (back to back): median {ts[len(ts)//2]:.1f} us = {canvas.numel() * 4 / ts[len(ts)//2] / 1e3:.0f} GB/s')
