"""round 6: one decoder-pipeline launch (2 scans x 256 rows x 32 steps) on the whole chip and on HALF of it (cu_budget = 128: 128-row groups forward, 64-row groups backward),
alone and beside the projection / state-gradient GEMM of the pipeline on another stream (gemm_nt_x6w_kernel).  HIP events; the pair is timed from the first launch to the join."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
torch.manual_seed(0)
src = open(os.path.join(R, "scratch", "r5_bench_fwd_scans.py")).read()
exec(src.split("def timeit")[0].split("torch.manual_seed(0)")[1])
fwd = mk(2, 256, 32, dense_every=2, h0=True)
srcb = open(os.path.join(R, "scratch", "r5_bench_bwd_scans.py")).read()
ns = {}
exec(srcb.split("def timeit")[0].split("torch.manual_seed(0)")[1].replace("def mk(", "def mkb("), globals(), ns)
bwd = ns["mkb"](2, 256, 32, False)
A1, W1, C1 = torch.randn(8192, 512, device=dev), torch.randn(1536, 512, device=dev) * 0.1, torch.empty(8192, 1536, device=dev)
A2, W2, C2 = torch.randn(8192, 1536, device=dev), torch.randn(512, 1536, device=dev) * 0.1, torch.empty(8192, 512, device=dev)
side = torch.cuda.Stream()
ops.dw_x6 = True


def run(kind, budget, with_gemm, reps=8, variant=0):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        if with_gemm:
            side.wait_stream(torch.cuda.current_stream())
        if kind == "fwd":
            ops.gru_seq_fwd(fwd, cu_budget=budget, variant=variant)
        else:
            ops.gru_seq_bwd(bwd, cu_budget=budget, variant=variant)
        if with_gemm:
            with torch.cuda.stream(side):
                ops.lane = "side"
                if kind == "fwd":
                    ops.gemm(A1, W1, C1, a_k=True, b_k=True)
                else:
                    ops.gemm(A2, W2, C2, a_k=True, b_k=True)
                ops.lane = ""
            torch.cuda.current_stream().wait_stream(side)
        e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    assert not ops.gru_sync_error()
    return min(ms[2:]) * 1e3, sum(ms[2:]) / (reps - 2) * 1e3


for kind in ("fwd", "bwd"):
    for budget, var, tag in ((0, 0, "whole chip"), (128, 0, "cu_budget 128"), (128, 0x10000, "cu_budget 128, XCDs 0-3")):
        for g in (False, True):
            best, mean = run(kind, budget, g, variant=var)
            print("%s launch, %-26s %-18s best %.1f us  mean %.1f us" % (kind, tag, "beside its GEMM" if g else "alone", best, mean), flush=True)
