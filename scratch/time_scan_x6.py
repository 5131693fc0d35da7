"""round 4: isolated timing of the opt-in bf16 x 6 forward scan (gru_fwd_x6_kernel) against the shipped ping-pong fp32 kernels: encoder shape (4 scans x 256 rows, T = 256,
128-row groups) and a 64-row-group shape (2 scans x 256 rows, T = 32, no initial state); us per launch and fp32-equivalent fraction of the fp32 MFMA peak"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
def mk(n, B, T):
    out = []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wf3 = torch.zeros(ops.frag_floats(3*H, H) * 3 // 2, device=dev); ops.frag3_pack(w, wf3)
        out.append(dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, w_hh_frag3=wf3, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                        gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                        h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev)))
    return out
def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
for name, n, B, T, reps in (("encoder 4 x 256 rows, T=256", 4, 256, 256, 10), ("2 x 256 rows (64-row groups), T=32", 2, 256, 32, 40)):
    sc = mk(n, B, T)
    flop = n * B * T * 2.0 * H * 3 * H
    for rnd in range(2):
        for x6 in (False, True):
            ops.dw_x6 = x6
            us = timeit(lambda: ops.gru_seq_fwd(sc), reps)
            print("%-40s bf16x6=%d: %8.1f us per launch = %6.2f us per step -> %.3f of the fp32 MFMA peak (fp32-equivalent)" % (name, x6, us, us / T, flop / (us * 1e-6) / 157.3e12), flush=True)
assert not ops.gru_sync_error()
