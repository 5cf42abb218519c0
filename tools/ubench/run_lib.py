"""run bench.py against another build of the library (A/B of kernel attributes): python tools/ubench/run_lib.py <lib.so> [bench args]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from monohair_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = ["bench.py"] + sys.argv[2:]
import bench
bench.main()
