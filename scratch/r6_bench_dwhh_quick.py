"""round 6: dW_hh 1536 x 512 x 65536 (16 K ranges) on the library named by FN_LIB (scratch/<name>), producer / consumer bf16 x 6 kernel only; HIP events"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"])
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev); H = 512
torch.manual_seed(0)
rows = int(os.environ.get("ROWS", "65536"))
dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
dW = torch.zeros(3 * H, H, device=dev)
ops.dw_x6 = True; ops.x6_wide = "force" if os.environ.get("WIDE", "1") == "1" else False
for sk in ((32,) if os.environ.get("WIDE", "1") == "1" else (16,)):
    ms = []
    for _ in range(12):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_dwhh(dgx, dghn, hp, dW, splitk=sk); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    print("%-16s %2d K ranges: best %.1f us mean %.1f us   checksum %.6e" % (os.environ.get("FN_LIB", "product"), sk, min(ms[2:]) * 1e3, sum(ms[2:]) / 10 * 1e3, float(dW.double().abs().sum())), flush=True)
