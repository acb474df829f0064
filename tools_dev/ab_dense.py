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
