#!/usr/bin/env python
"""Target for the rocprofv3 --pmc passes: the product scatter (voxelizer-style unique coordinates)
at BASELINE config #2, 6 launches per layout."""
import os
import sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch
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
for cl in (True, False):
    for _ in range(6):
        y = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=cl, unique=True)
        torch.cuda.synchronize()
        del y
print('algorithmic bytes', B * M * Cc * 4 + B * M * 16 + B * Cc * ny * nx * 4)
