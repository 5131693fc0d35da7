"""forward step cost with table gather (token -> row, a dependent load chain) vs dense pre-activations vs neither"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, H, V = 256, 64, 512, 342
def mk(n, mode, gates=True):
    fw = []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 h0=torch.randn(B, H, device=dev) * 0.1, h_all=torch.zeros(T, B, H, device=dev),
                 gates=torch.zeros(T, ops.gates_floats(B, H), device=dev) if gates else None)
        if mode == "table":
            d["gx_table"] = torch.randn(V, 3*H, device=dev) * 0.1; d["idx"] = torch.randint(0, V, (B, T), dtype=torch.int32, device=dev)
        elif mode == "dense":
            d["gx_dense"] = torch.randn(T, B, 3*H, device=dev) * 0.1
        fw.append(d)
    return fw
def timeit(fn, n_launch, reps=3):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); best = min(best, e0.elapsed_time(e1) * 1e3 / n_launch)
    return best
for n in (1, 4):
    for mode in ("table", "dense", "none"):
        for gates in (True, False):
            fw = mk(n, mode, gates)
            print("scans=%d gx=%-5s gates=%-5s %7.2f us/launch" % (n, mode, gates, timeit(lambda: ops.gru_seq_fwd(fw), T)), flush=True)
