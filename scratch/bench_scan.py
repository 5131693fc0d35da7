"""Micro-benchmark of the GRU step kernels: us per launch for the encoder-like (4 scans) and decoder-like (1 scan) cases,
forward and backward, over tile / prefetch configurations (FN_*_CFG env overrides)."""
import os, sys, itertools
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
B, T, H, V = 256, 64, 512, 342
torch.manual_seed(0)
def mk(n):
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 h0=torch.randn(B, H, device=dev) * 0.1, gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        fw.append(d)
        bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d["h0"], h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, B, H, device=dev) * 0.01,
                       dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), dh0=torch.zeros(B, H, device=dev),
                       dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev), scratch=torch.zeros(B, H, device=dev)))
    return fw, bw
def timeit(fn, n_launch, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / n_launch)
    return best
fw4, bw4 = mk(4); fw1, bw1 = mk(1)
def run(var, cfgs, fn, nl, label):
    for c in cfgs:
        os.environ[var] = c
        try:
            t = timeit(fn, nl)
            print("%-14s %-8s %8.2f us/launch" % (label, c, t), flush=True)
        except Exception as e:
            print(label, c, "ERR", str(e)[:80])
    os.environ.pop(var, None)
fcfgs = ["1,0,2", "1,0,3", "1,0,4", "2,0,2", "2,0,3", "2,0,4", "4,0,1", "4,0,2", "4,0,3"]
bcfgs = ["1,1,3", "1,1,4", "1,2,3", "1,2,4", "2,1,2", "2,1,3", "2,1,4", "2,2,2", "2,2,3", "2,2,4", "4,1,2", "4,1,3", "4,2,2", "4,2,3", "4,2,4"]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "fwd4only":
    c = os.environ.get("FN_FWD_CFG", "default")
    print("fwd 4 scans", c, timeit(lambda: ops.gru_seq_fwd(fw4), T, reps=2))
if which in ("all", "fwd"):
    run("FN_FWD_CFG", fcfgs, lambda: ops.gru_seq_fwd(fw4), T, "fwd 4 scans")
    run("FN_FWD_CFG1", fcfgs, lambda: ops.gru_seq_fwd(fw1), T, "fwd 1 scan")
if which in ("all", "bwd"):
    run("FN_BWD_CFG", bcfgs, lambda: ops.gru_seq_bwd(bw4), T + 1, "bwd 4 scans")
    run("FN_BWD_CFG1", bcfgs, lambda: ops.gru_seq_bwd(bw1), T + 1, "bwd 1 scan")
