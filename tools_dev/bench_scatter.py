#!/usr/bin/env python
"""Dev micro-benchmark of the pillar-scatter kernels at BASELINE config #2
([256000,64] -> [16,64,496,432]). `GGA_SCATTER_VARIANT` selects the NCHW kernel."""
import ctypes as C
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def one():
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
    algo = B * M * Cc * 4 + B * M * 16 + B * Cc * ny * nx * 4
    L = _lib.lib()
    out = {}
    for name, layout in (('nchw', 0), ('nhwc', 1)):
        canvas = torch.empty(B * Cc * ny * nx, device=dev)
        cmap = F._cell_map(dev, B, ny, nx)
        a, b_ = C.c_float(0), C.c_float(0)
        args = (F._p(feats), F._p(coors), B * M, B, Cc, ny, nx, layout, F._p(cmap), F._p(canvas))
        _lib.check(L.gga_profile_pillar_scatter(*args, 5, C.byref(a), C.byref(b_), F._stream()), 'p')
        _lib.check(L.gga_profile_pillar_scatter(*args, 30, C.byref(a), C.byref(b_), F._stream()), 'p')
        out[name] = (a.value, b_.value, algo / b_.value / 1e6)
        if os.environ.get('GGA_SCATTER_VARIANT') != '9':
            ref = F.pillar_scatter(feats, coors, B, ny, nx, channels_last=bool(layout))
            idx = coors.long()
            assert torch.equal(ref[idx[:, 0], :, idx[:, 2], idx[:, 3]], feats)
            assert int((ref != 0).sum()) == int((feats != 0).sum())
    # reference points: hipMemset of the canvas and a torch fill
    canvas = torch.empty(B * Cc * ny * nx, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        canvas.zero_()
    e0.record()
    for _ in range(20):
        canvas.zero_()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"variant={os.environ.get('GGA_SCATTER_VARIANT', 'default')} "
          + ' '.join(f'{k}: map {v[0]*1e3:.1f}us canvas {v[1]*1e3:.1f}us {v[2]:.0f} GB/s' for k, v in out.items())
          + f' | torch zero_ {ms*1e3:.1f}us {canvas.numel()*4/ms/1e6:.0f} GB/s', flush=True)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'one':
        one()
    else:
        for v in sys.argv[1:] or ['0', '1', '2', '9']:
            if v.startswith('u'):
                subprocess.call([sys.executable, __file__, 'one'], env=dict(os.environ, GGA_SCATTER_NHWC_UNROLL=v[1:]))
            else:
                subprocess.call([sys.executable, __file__, 'one'], env=dict(os.environ, GGA_SCATTER_VARIANT=v))
