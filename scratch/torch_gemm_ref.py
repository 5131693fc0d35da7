"""What do the ROCm libraries (rocBLAS / hipBLASLt through torch.mm) reach on this path's big fp32 GEMM shapes?"""
import torch
dev = torch.device("cuda:0")
torch.backends.cuda.matmul.allow_tf32 = False
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1))
    return best
K = 65280
A = torch.randn(K, 1536, device=dev); B = torch.randn(K, 512, device=dev)
X = torch.randn(65536, 512, device=dev); W = torch.randn(1536, 512, device=dev); G = torch.randn(65536, 1536, device=dev)
for name, fn, fl in (("dW  = A^T B   [1536x512], K=65280", lambda: torch.mm(A.t(), B), 2.0 * 1536 * 512 * K),
                     ("gx2 = X W^T   [65536x1536], K=512", lambda: torch.mm(X, W.t()), 2.0 * 65536 * 1536 * 512),
                     ("dX  = G W     [65536x512], K=1536", lambda: torch.mm(G, W), 2.0 * 65536 * 512 * 1536)):
    ms = t(fn)
    print("%-40s %8.1f us  %6.1f TFLOP/s" % (name, ms * 1e3, fl / ms / 1e9))
import os
for pref in ("hipblaslt", "hipblas"):
    try:
        torch.backends.cuda.preferred_blas_library(pref)
        ms = t(lambda: torch.mm(A.t(), B)); print("preferred %s: dW %8.1f us %6.1f TFLOP/s" % (pref, ms * 1e3, 2.0 * 1536 * 512 * K / ms / 1e9))
    except Exception as e:
        print(pref, "n/a", str(e)[:80])
