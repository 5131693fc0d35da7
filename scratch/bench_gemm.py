import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import shutil
if len(sys.argv) > 1:          # A/B of two builds inside one gpurun call: python bench_gemm.py scratch/lib_x.so
    shutil.copy(os.path.join(R, sys.argv[1]), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
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
cases = [("dW all sk16", False, False, 1536, 512, 65280, 16), ("dW out sk42", False, False, 342, 512, 65536, 42),
         ("fwd proj32", True, True, 8192, 1536, 512, 1), ("fwd proj64", True, True, 16384, 1536, 512, 1), ("logits", True, True, 65536, 342, 512, 1),
         ("dX out", True, False, 65536, 512, 342, 1), ("dX proj32", True, False, 8192, 512, 1536, 1), ("dX proj32 NT", True, True, 8192, 512, 1536, 1), ("dX out NT352", True, True, 65536, 512, 352, 1), ("dX proj64", True, False, 16384, 512, 1536, 1)]
for name, ak, bk, M, N, K, sk in cases:
    lda = 344 if (not ak and M == 342) else (M if not ak else (344 if K == 342 else K))
    A = torch.randn((K, lda) if not ak else (M, lda), device=dev)
    Av = A[:, :M] if not ak else A[:, :K]
    Bm = torch.randn((N, K) if bk else (K, N), device=dev)
    ldc = 344 if N == 342 else N
    C = torch.zeros(M, ldc, device=dev)
    ms = t(lambda: ops.gemm(Av, Bm, C[:, :N], a_k=ak, b_k=bk, splitk=sk))
    extra = ""
    if ak and bk and sk == 1:          # the <= 128-register instance of the LDS-free NT kernel (4 workgroups per CU)
        ml = t(lambda: ops.gemm(Av, Bm, C[:, :N], a_k=ak, b_k=bk, splitk=sk, lean=True))
        extra = "   lean %8.1f us  %6.1f TFLOP/s" % (ml * 1e3, 2.0 * M * N * K / ml / 1e9)
    print("%-14s M=%6d N=%5d K=%6d splitk=%2d  %8.1f us  %6.1f TFLOP/s%s" % (name, M, N, K, sk, ms * 1e3, 2.0 * M * N * K / ms / 1e9, extra), flush=True)
