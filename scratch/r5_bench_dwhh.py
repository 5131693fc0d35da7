"""round 5: the weight-gradient product of a scan (dW_hh = [dgx[:, :2H] | dghn]^T h, 1536 x 512 x 65280, 16 K ranges) and of the output layer (342 x 512 x 65536,
42 ranges) on the fp32 MFMA, the bf16 x 6 kernel with per-wave splits and the one that shares the split through LDS; HIP events, each launch alone"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"]); print("library:", _lib.LIB_PATH)
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H = 512
torch.manual_seed(0)
rows = 65280
dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
dW = torch.zeros(3 * H, H, device=dev)
dl, hx = torch.randn(65536, 352, device=dev), torch.randn(65536, H, device=dev)
dWo = torch.zeros(342, H, device=dev)


def t(fn, reps=8):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms[2:]) * 1e3, sum(ms[2:]) / (reps - 2) * 1e3


for sk in (16, 21, 32):
    ref = None
    for tag, x6, priv in (("fp32 MFMA", False, False), ("bf16x6", True, False)):
        ops.dw_x6 = x6
        best, mean = t(lambda: ops.gru_dwhh(dgx, dghn, hp, dW, splitk=sk))
        if ref is None:
            ref = dW.clone()
        print("dW_hh 1536x512x%d, %2d K ranges  %-28s best %.1f us mean %.1f us = %.1f fp32-equivalent TFLOP/s   max diff vs fp32 %.2e" %
              (rows, sk, tag, best, mean, 2.0 * rows * 3 * H * H / best / 1e6, float((dW - ref).abs().max()) / float(ref.abs().max())), flush=True)
for sk in (42, 32):
    for tag, x6, priv in (("fp32 MFMA", False, False), ("bf16x6", True, False)):
        ops.dw_x6 = x6
        best, mean = t(lambda: ops.gemm(dl[:, :342], hx, dWo, a_k=False, b_k=False, splitk=sk))
        print("dW_out 342x512x65536, %2d K ranges  %-28s best %.1f us mean %.1f us" % (sk, tag, best, mean), flush=True)
ops.dw_x6 = False
