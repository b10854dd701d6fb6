"""Run a tool against an experiment build of the library: with_lib.py <lib.so> <script.py> [args ...]"""
import runpy
import sys
sys.path.insert(0, '.')
from normalisr_amd import _lib
_lib.LIB_PATH = sys.argv[1]
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
