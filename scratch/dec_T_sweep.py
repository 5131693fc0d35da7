"""round 4: decoder-shaped weight-stationary launches (2 scans x 256 rows) at T = 8 .. 128 - us per launch = a + b T: the fixed cost of a
launch (prologue: weight slice to LDS / registers, first ring, last flush) against the cost of a time step.  Shipped kernels (variant 0)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
exec(open(os.path.join(R, "scratch", "pp_time.py")).read().split("def mk(")[1].join(["def mk(", ""]).split("for name, n, B, T, reps in")[0]) if False else None

def mk(n, B, T):
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=B, T=T, H=H, reverse=0, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev), h0=torch.randn(B, H, device=dev) * 0.1)
        fw.append(d)
        bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d["h0"], h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, B, H, device=dev) * 0.01,
                       dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev),
                       dh0=torch.zeros(B, H, device=dev), dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev)))
    return fw, bw

def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

Ts = (8, 16, 32, 64, 128)
res = {"fwd": [], "bwd": []}
for T in Ts:
    fw, bw = mk(2, 256, T)
    res["fwd"].append(min(timeit(lambda: ops.gru_seq_fwd(fw), 40) for _ in range(3)))
    ops.gru_seq_fwd(fw)
    res["bwd"].append(min(timeit(lambda: ops.gru_seq_bwd(bw), 40) for _ in range(3)))
for k, us in res.items():
    b, a = np.polyfit(np.array(Ts, float), np.array(us), 1)
    print("%s: us per launch at T = %s: %s   ->  fit %.1f us + %.2f us x T  (MFMA minimum per step at 2.4 GHz: 5.1 us)" % (k, Ts, " ".join("%.1f" % u for u in us), a, b))
assert not ops.gru_sync_error()
