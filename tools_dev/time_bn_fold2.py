"""How long does bn_fwd_final_kernel (the BatchNorm fold over a producer's f64 partial rows) take as a function of the number of
rows and channels? Back-to-back launches, HIP events. Usage: python tools_dev/time_bn_fold2.py"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import _lib, functional as F
L = _lib.lib()
dev = torch.device('cuda:0')
rows = 16 * 248 * 216
for C in (64, 128, 256):
    for tiles in (64, 256, 1024, 2048, 3472):
        st = torch.rand(tiles, 2, C, dtype=torch.float64, device=dev)
        st[:, 1] += 2.0
        gam, bet = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        saved, ss = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev)
        def go():
            L.gga_bn_stats_partials(F._p(gam), F._p(bet), F._p(rm), F._p(rv), rows, C, 1e-3, 0.01, F._p(saved), F._p(ss), F._p(st), tiles, F._stream())
        for _ in range(20):
            go()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(200):
            go()
        b.record()
        torch.cuda.synchronize()
        print(f'C {C:3d} rows {tiles:4d}: {a.elapsed_time(b) / 200 * 1e3:6.1f} us per launch (back to back, incl. the launch boundary)')
