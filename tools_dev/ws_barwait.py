"""Who waits for whom at the stage barriers of dense_conv3x3_ws_kernel? (stamped build: s_memtime around every stage barrier of
consumer wave 0, halo wave 4 and the weight wave; tools_dev/exp_libs/libgga_wsbarwait.so)
    python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_wsbarwait.so tools_dev/ws_barwait.py
Cycles per stage spent inside the barrier by each role (the role that arrives last waits least)."""
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
    tiles_per_wg = np.ceil((B * -(-H // (8 if cout == 128 else 16)) * -(-W // 32) - np.arange(256)) / 256)
    stages = (tiles_per_wg * 9 * cin // 16).sum()
    print(f'{B}x{cin}->{cout}x{H}x{W}: cycles per stage inside the barrier: consumer {s[:, 0].sum() / stages:.0f}, halo wave {s[:, 1].sum() / stages:.0f}, '
          f'weight wave {s[:, 2].sum() / stages:.0f}; consumer total per stage {s[:, 3].sum() / stages:.0f}')
