"""Same-box A/B of the dense-convolution routing: `python tools_dev/ab_dense.py MODE` runs bench.py
(--no-cpu-baseline) with MODE in {full, nowide (256-channel convs back on MIOpen), notr (no
transposed tile walk), nowide_notr}. Boxes differ by a few per cent, so alternate the modes in one
gpurun call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import runpy
import gga_amd.dense_conv as d
mode = sys.argv[1]
if 'nowide' in mode:
    _el = d.eligible
    d.eligible = lambda conv, x: _el(conv, x) and conv.out_channels <= 128 and conv.in_channels != 256
if 'notr' in mode:
    d._transposed = lambda H, W: False
    d._cdiv_orig = d._cdiv
sys.argv = ['bench.py', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
