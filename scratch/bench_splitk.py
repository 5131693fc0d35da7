"""split-K sweep of the TN weight-gradient GEMM for the attribute decoders' dW_hh (K = Tr*B - B = 16128) and the output layer's dW (342 x 512, K = 65536)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
def t(fn, reps=8):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
for name, M, N, K, lda, sks in (("attr dW_hh", 1536, 512, 16128, 1536, (4, 5, 6, 8, 10, 12, 16, 24)), ("dW out", 342, 512, 65536, 344, (16, 24, 32, 40, 42, 48, 56, 64, 80)),
                                ("dW_hh 65280", 1536, 512, 65280, 1536, (8, 16, 24, 32))):
    A = torch.randn(K, lda, device=dev); Bm = torch.randn(K, N, device=dev); C = torch.zeros(M, N, device=dev)
    for sk in sks:
        ms = t(lambda: ops.gemm(A[:, :M], Bm, C, a_k=False, b_k=False, splitk=sk))
        print("%-12s M=%5d N=%4d K=%6d splitk=%3d  %7.1f us  %6.1f TFLOP/s" % (name, M, N, K, sk, ms * 1e3, 2.0 * M * N * K / ms / 1e9), flush=True)
