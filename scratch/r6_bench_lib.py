"""bench.py on the library named by FN_LIB (scratch/<name>; A/B measurements of library variants in one session)"""
import os, runpy, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"])
sys.argv[0] = os.path.join(R, "bench.py")
runpy.run_path(sys.argv[0], run_name="__main__")
