"""Cycle accounting of dense_conv3x3_x9_kernel's stage loop (a -DDC_PROBE build of the library, see sparse_conv.hip):
per (tap, chunk) stage of wave 0 - fragment reads until the data is there, MFMA issue, weight store, barrier wait; per
chunk - halo split + store and its barrier.   python tools_dev/run_with_lib.py tools_dev/exp_libs/libexp_probe.so tools_dev/probe_dense_stage.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import torch
from gga_amd import _lib, dense_conv
L = _lib.lib()
DEV = 'cuda:0'
for cin, cout, B, H, W in ((64, 64, 16, 248, 216), (128, 128, 16, 124, 108), (960, 64, 16, 248, 216)):
    conv = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=False).to(DEV).to(memory_format=torch.channels_last)
    x = torch.randn(B, cin, H, W, device=DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        for _ in range(3):
            dense_conv.conv2d(x, conv)
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 8)()
        L.gga_debug_dc_probe(buf)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        dense_conv.conv2d(x, conv)
        e1.record()
        torch.cuda.synchronize()
        L.gga_debug_dc_probe(buf)
    v = list(buf)
    stages = max(v[4], 1)
    tot = v[7]
    print(f'{cin}->{cout} {B}x{H}x{W}: op {e0.elapsed_time(e1) * 1e3:.0f} us; per stage of wave 0 (cycles): fragment reads {v[0] / stages:.0f}, '
          f'MFMA issue {v[1] / stages:.0f}, weight store {v[2] / stages:.0f}, barrier {v[3] / stages:.0f}; per chunk: halo store {9 * v[5] / stages:.0f}, '
          f'its barrier {9 * v[6] / stages:.0f}; accounted {(v[0] + v[1] + v[2] + v[3] + v[5] + v[6]) / tot:.2f} of the kernel\'s {tot / max(1, v[4]) * 1.0:.0f} cycles per stage')
