"""round 5: in-kernel stamps of the bf16 x 6 backward scan (scratch/lib_timing.so = -DFN_TIMING build): per phase [start, K loop done, barrier passed, epilogue done]
for halves A and B of iteration 10 of workgroups 0-7, and how often workgroup 0 found the other half's counter short (poll fallback)."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch, numpy as np
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
_lib.LIB_PATH = os.path.join(R, "scratch", os.environ.get("FN_LIB", "lib_timing.so"))
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
lib = ops.lib
H = 512
torch.manual_seed(0)
exec(open(os.path.join(R, "scratch", "r5_bench_bwd_scans.py")).read().split("def timeit")[0].split("torch.manual_seed(0)")[1])
buf = (ctypes.c_ulonglong * 64)()
cnt = (ctypes.c_ulonglong * 8)()
lib.fn_pdbg_read.argtypes = [ctypes.c_void_p]
lib.fn_pcnt_read.argtypes = [ctypes.c_void_p]
for name, scans, T in (("encoder 4 x 256 x 64 steps", mk(4, 256, 64, True), 64), ("decoder 2 x 256 x 32 steps", mk(2, 256, 32, False), 32)):
    for tag, x6 in (("fp32 rs", False), ("bf16x6", True)):
        ops.dw_x6 = x6
        for rep in range(3):
            lib.fn_pcnt_read(cnt); c0 = cnt[0]
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.gru_seq_bwd(scans); e1.record(); torch.cuda.synchronize()
            lib.fn_pdbg_read(buf); lib.fn_pcnt_read(cnt)
            a = np.array(list(buf), dtype=np.int64).reshape(8, 8)
            d = a - a[:, :1]
            print("%s | %s rep %d: %.1f us per step; poll fallbacks of workgroup 0: %d of %d phases" % (name, tag, rep, e0.elapsed_time(e1) * 1e3 / T, cnt[0] - c0, 2 * T))
            if rep == 2:
                for r in d[:3]:
                    print("     stamps of iteration 10 (ticks since phase A start) A[start kloop barrier epi] B[...]:", r.tolist())
        ops.dw_x6 = False
