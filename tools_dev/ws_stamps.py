"""In-kernel clocks of dense_conv3x3_ws_kernel's consumers (experiment build with s_memtime stamps around the main loop and
the epilogue of every tile; tools_dev/exp_libs/libgga_wsstamps.so, built from a stamped copy of dense_conv_ws.hip - round 5):
    python tools_dev/run_with_lib.py tools_dev/exp_libs/libgga_wsstamps.so tools_dev/ws_stamps.py
Per shape: share of consumer wave 0's time in the main loop / the epilogue / elsewhere, shader cycles per stage."""
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
    for data in ('random', 'zeros'):
        x = (torch.randn(B, cin, H, W, device=dev) if data == 'random' else torch.zeros(B, cin, H, W, device=dev)).contiguous(memory_format=torch.channels_last)
        for stats in (True, False):
            for _ in range(20):
                dense_conv._run(x, w, False, stats)
            torch.cuda.synchronize()
            out = np.zeros(1024, np.uint64)
            assert L.gga_debug_ws_stamps(out.ctypes.data) == 0
            s = out.reshape(256, 4).astype(np.float64)
            main, epi, tiles, total = s[:, 0].sum(), s[:, 1].sum(), s[:, 2].sum(), s[:, 3].sum()
            stages = 9 * cin // 16
            print(f'{B}x{cin}->{cout}x{H}x{W} {data:6s} stats {int(stats)}: tiles/WG {tiles / 256:.2f}  main loop {main / total:.3f}  epilogue {epi / total:.3f}  '
                  f'rest {1 - (main + epi) / total:.3f}  | per tile: main {main / tiles:.0f} cycles ({main / tiles / stages:.1f} per stage; the matrix instructions alone: {24 * 32}), '
                  f'epilogue {epi / tiles:.0f}; per WG {total / 256:.0f}')
