import os, sys, shutil
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
v = os.environ.get("KNOCK", "")
if v:
    shutil.copy(os.path.join(R, "scratch/lib_knock_%s.so" % v), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
sys.argv = [sys.argv[0], "x4only"]
exec(open(os.path.join(R, "scratch/test_persist.py")).read())
