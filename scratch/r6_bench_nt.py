"""round 6: the decoder pipeline's Linear-forward / dX products on the fp32 MFMA (gemm_nt_direct_kernel) and on the bf16 x 6 producer / consumer kernel; HIP events"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
torch.manual_seed(0)
for tag, M, N, K in (("gx2 = hx0 W_ih2^T (chunk)", 8192, 1536, 512), ("dhx0 = dgx2 W_ih2 (chunk)", 8192, 512, 1536), ("dhx1 = dlogits W_out", 65536, 512, 352)):
    A, W, C = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.1, torch.empty(M, N, device=dev)
    for x6, per_tile in ((False, False), (True, True), (True, False)):
        ops.dw_x6, ops.nt_x6, ops.x6_per_tile = x6, x6, per_tile
        ms = []
        for _ in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gemm(A, W, C, a_k=True, b_k=True); e1.record(); torch.cuda.synchronize()
            ms.append(e0.elapsed_time(e1))
        best = min(ms[2:]) * 1e3
        print("%-28s %5d x %4d x %4d  %-11s best %.1f us mean %.1f us = %.1f fp32-equivalent TFLOP/s" % (tag, M, N, K, ("bf16x6/tile" if per_tile else "bf16x6/cu") if x6 else "fp32", best, sum(ms[2:]) / 10 * 1e3, 2.0 * M * N * K / best / 1e6), flush=True)
