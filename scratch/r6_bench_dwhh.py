"""round 6: the weight-gradient products on the fp32 MFMA, the round-5 bf16 x 6 kernel (every wave splits its own operands) and the producer / consumer
kernel (gemm_tn_x6w_kernel); HIP events, each launch alone.  Also checks that the two bf16 x 6 kernels agree bit for bit."""
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


def t(fn, reps=10):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return min(ms[2:]) * 1e3, sum(ms[2:]) / (reps - 2) * 1e3


MODES = (("fp32 MFMA", False, False, False), ("bf16x6 per-wave splits (r5)", True, True, False), ("bf16x6 producer/consumer 128x128", True, False, False), ("bf16x6 producer/consumer 128x256", True, False, "force"))
for rows in (65280, 65536, 16384):
    dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
    dW = torch.zeros(3 * H, H, device=dev)
    for sk in (16, 32, 48):
        ref, prev = None, None
        for tag, x6, pw, wide in MODES:
            ops.dw_x6, ops.x6_perwave, ops.x6_wide = x6, pw, wide
            best, mean = t(lambda: ops.gru_dwhh(dgx, dghn, hp, dW, splitk=sk))
            if ref is None:
                ref = dW.clone()
            same = ""
            if x6:
                if prev is not None:
                    same = "  bit-identical to per-wave: %s" % bool(torch.equal(prev, dW))
                prev = dW.clone()
            print("dW_hh 1536x512x%d, %2d K ranges  %-36s best %.1f us mean %.1f us = %.1f fp32-equivalent TFLOP/s   max diff vs fp32 %.2e%s" %
                  (rows, sk, tag, best, mean, 2.0 * rows * 3 * H * H / best / 1e6, float((dW - ref).abs().max()) / float(ref.abs().max()), same), flush=True)
    del dgx, dghn, hp
dl, hx = torch.randn(65536, 352, device=dev), torch.randn(65536, H, device=dev)
dWo = torch.zeros(342, H, device=dev)
for sk in (42, 32, 64):
    prev = None
    for tag, x6, pw, wide in MODES:
        ops.dw_x6, ops.x6_perwave, ops.x6_wide = x6, pw, wide
        best, mean = t(lambda: ops.gemm(dl[:, :342], hx, dWo, a_k=False, b_k=False, splitk=sk))
        same = ""
        if x6:
            if prev is not None:
                same = "  bit-identical to per-wave: %s" % bool(torch.equal(prev, dWo))
            prev = dWo.clone()
        print("dW_out 342x512x65536, %2d K ranges  %-36s best %.1f us mean %.1f us%s" % (sk, tag, best, mean, same), flush=True)
# dense dW_ih2: 1536 x 512 x 65536 through fn_gemm_f32
dg, hx0 = torch.randn(65536, 3 * H, device=dev), torch.randn(65536, H, device=dev)
dWi = torch.zeros(3 * H, H, device=dev)
for sk in (16, 24):
    prev = None
    for tag, x6, pw, wide in MODES:
        ops.dw_x6, ops.x6_perwave, ops.x6_wide = x6, pw, wide
        best, mean = t(lambda: ops.gemm(dg, hx0, dWi, a_k=False, b_k=False, splitk=sk))
        same = ""
        if x6:
            if prev is not None:
                same = "  bit-identical to per-wave: %s" % bool(torch.equal(prev, dWi))
            prev = dWi.clone()
        print("dW_ih2 1536x512x65536, %2d K ranges  %-36s best %.1f us mean %.1f us = %.1f TFLOP/s%s" % (sk, tag, best, mean, 2.0 * 65536 * 3 * H * H / best / 1e6, same), flush=True)
ops.dw_x6, ops.x6_perwave, ops.x6_wide = False, False, True
