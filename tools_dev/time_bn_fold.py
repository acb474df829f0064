"""BatchNorm statistics + backward of a 64-channel 16 x 248 x 216 map: time of the whole calls (the fold kernels are their tails)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gga_amd import functional as F, _lib
L = _lib.lib()
dev = 'cuda:0'
for C, H, W in ((64, 248, 216), (128, 124, 108), (256, 62, 54)):
    B = 16
    rows = B * H * W
    x = torch.randn(B, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    gamma, beta = torch.rand(C, device=dev) + 0.5, torch.rand(C, device=dev) - 0.5
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    saved, ss = torch.empty(2 * C, device=dev), torch.empty(2 * C, device=dev)
    ws = torch.empty(L.gga_bn_relu_workspace_bytes(rows, C), dtype=torch.uint8, device=dev)
    def stats():
        _lib.check(L.gga_bn_stats(F._p(x), F._p(gamma), F._p(beta), F._p(rm), F._p(rv), rows, C, 1e-3, 0.01, 1, F._p(saved), F._p(ss), F._p(ws), ws.numel(), F._stream()), 'stats')
    for _ in range(3): stats()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): stats()
    e1.record(); torch.cuda.synchronize()
    print(os.path.basename(_lib.LIB_PATH), (C, H, W), 'gga_bn_stats %.1f us' % (e0.elapsed_time(e1) / 20 * 1e3), flush=True)
