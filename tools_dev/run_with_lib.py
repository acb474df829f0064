"""`python tools_dev/run_with_lib.py <libgga_hip variant .so> <script.py> [args...]`: run a script with an
experimental build of the library (timing experiments; variants live under tools_dev/exp_libs/)."""
import os, runpy, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from gga_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
