"""round 5: forward scans at the two launch shapes of the training step - encoder (4 scans x 256 rows x 256 steps, 128-row groups) and one decoder-pipeline
launch (2 scans x 256 rows x 32 steps, 64-row groups, initial state + hand-over image) - on the fp32 MFMA (ping-pong), the bf16 x 6 single-group kernel
(variant bit 15) and the bf16 x 6 ping-pong kernel; HIP events on the launch stream, each launch alone.  Optional argv[1]: FN_TIMING counter read-out."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd import _lib
if os.environ.get("FN_LIB"):                       # A/B of library builds (scratch/r5_build_variant.sh)
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"])
    print("library:", _lib.LIB_PATH)
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
torch.manual_seed(0)


def mk(n, B, T, dense_every=0, h0=False):
    out = []
    for s in range(n):
        w = (torch.randn(3 * H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3 * H, H), device=dev); ops.frag_pack(w, wf)
        wf3 = torch.zeros(ops.frag_floats(3 * H, H) * 3 // 2, device=dev); ops.frag3_pack(w, wf3)
        d = dict(B=B, T=T, H=H, reverse=s & 1 if n == 4 else 0, w_hh_frag=wf, w_hh_frag3=wf3, b_hh=torch.zeros(3 * H, device=dev), b_ih=torch.zeros(3 * H, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        if dense_every and s % dense_every == dense_every - 1:
            d["gx_dense"] = torch.randn(T, B, 3 * H, device=dev) * 0.1
        else:
            d["gx_table"] = torch.randn(V, 3 * H, device=dev) * 0.1
            d["idx"] = torch.randint(0, V, (B, T), dtype=torch.int32, device=dev)
        if h0:
            d["h0"] = torch.randn(B, H, device=dev) * 0.1
        out.append(d)
    return out


def timeit(scans, x6, variant, reps=6):
    ops.dw_x6, ops.variant = x6, variant
    ms = []
    for rep in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.gru_seq_fwd(scans); e1.record(); torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ops.dw_x6, ops.variant = False, 0
    assert not ops.gru_sync_error()
    return min(ms[1:]), sum(ms[1:]) / (reps - 1)


for name, scans, T in (("encoder 4 x 256 rows x 256 steps", mk(4, 256, 256), 256), ("decoder launch 2 x 256 rows x 32 steps", mk(2, 256, 32, dense_every=2, h0=True), 32),
                       ("decoder launch 2 x 256 rows x 64 steps", mk(2, 256, 64, dense_every=2, h0=True), 64)):
    ref = None
    for tag, x6, var in (("fp32 ping-pong", False, 0), ("bf16x6 single-group", True, 0x8000), ("bf16x6 ping-pong", True, 0)):
        best, mean = timeit(scans, x6, var)
        h = scans[0]["h_all"].clone()
        if ref is None:
            ref = h
        print("%-42s %-20s best %.3f ms  mean %.3f ms  = %.2f us per step   max |h - h_fp32| = %.2e" % (name, tag, best, mean, best * 1e3 / T, float((h - ref).abs().max())), flush=True)
