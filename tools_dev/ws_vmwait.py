"""How long does a halo wave of dense_conv3x3_ws_kernel wait for the piece it requested a chunk ago? (stamped build
tools_dev/exp_libs/libgga_wsvmwait.so: s_memtime around an explicit s_waitcnt vmcnt(NA - 1) in front of every piece's split)"""
import ctypes as C
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from gga_amd import _lib, dense_conv

dense_conv.PLANES = 2
os.environ['GGA_DC_WS_MFMA'] = '32'
L = _lib.lib()
L.gga_debug_ws_stamps.restype = C.c_int
L.gga_debug_ws_stamps.argtypes = [C.c_void_p]
dev = 'cuda:0'
for B, cin, cout, H, W in [(16, 128, 128, 124, 108), (16, 64, 64, 248, 216), (16, 384, 64, 248, 216)]:
    w = (torch.randn(cout, cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    for _ in range(20):
        dense_conv._run(x, w, False, True)
    torch.cuda.synchronize()
    out = np.zeros(2048, np.uint64)
    assert L.gga_debug_ws_stamps(out.ctypes.data) == 0
    s = out.reshape(256, 8).astype(np.float64)
    print(f'{B}x{cin}->{cout}x{H}x{W}: halo wave 4: {s[:, 0].sum() / s[:, 1].sum():.0f} cycles per piece waiting for its data ({s[:, 0].sum() / s[:, 2].sum():.3f} of the wave\'s time; '
          f'{s[:, 1].sum() / 256:.0f} pieces per workgroup lane)')
