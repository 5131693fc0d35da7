"""Weight-stationary single-launch scans vs the per-step kernels: equality of h_all / gates and time per step."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
V = 342
torch.manual_seed(0)
def mk(n, B, T, H, kind="table", h0=True, Ts=None):
    fw = []
    for s in range(n):
        Tq = Ts[s] if Ts else T
        w = (torch.randn(3*H, H, device=dev) / (H ** 0.5)).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        d = dict(B=B, T=Tq, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, b_ih=torch.randn(3*H, device=dev) * 0.1,
                 h_all=torch.zeros(Tq, B, H, device=dev), gates=torch.zeros(Tq, ops.gates_floats(B, H), device=dev))
        if h0: d["h0"] = torch.randn(B, H, device=dev) * 0.3
        if kind in ("table", "both"):
            d["gx_table"] = torch.randn(V, 3*H, device=dev) * 0.3
            d["idx"] = torch.randint(0, V, (B, Tq), dtype=torch.int32, device=dev)
            if s == 2: d["idx_shift"], d["start_token"] = -1, V - 1
        if kind in ("dense", "both"):
            d["gx_dense"] = torch.randn(Tq, B, 3*H, device=dev) * 0.3
        if s % 2 == 1: d["gx_rowbias"] = torch.randn(B, 3*H, device=dev) * 0.2
        fw.append(d)
    return fw
def run(fw, persistent, **kw):
    for d in fw: d["h_all"].fill_(float("nan")); d["gates"].zero_()
    ops.gru_seq_fwd(fw, persistent=persistent, **kw)
    torch.cuda.synchronize()
    return [(d["h_all"].clone(), d["gates"].clone()) for d in fw]
def timeit(fn, nsteps, reps=3):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / nsteps)
    return best
cases = [("small H64 x4 scans", dict(n=4, B=6, T=20, H=64, kind="table")),
         ("small H64 dense no h0", dict(n=1, B=37, T=9, H=64, kind="dense", h0=False)),
         ("H96 x3 ragged T", dict(n=3, B=21, T=12, H=96, kind="both", Ts=[12, 5, 12])),
         ("H512 B100 x1", dict(n=1, B=100, T=16, H=512, kind="table")),
         ("H512 B256 x1", dict(n=1, B=256, T=32, H=512, kind="dense")),
         ("H512 B256 x4", dict(n=4, B=256, T=64, H=512, kind="table")),
         ("H512 B256 x3 ragged", dict(n=3, B=256, T=32, H=512, kind="table", Ts=[32, 8, 8]))]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which == "x4only": cases = [c for c in cases if c[0] == "H512 B256 x4"]
for name, kw in cases:
    fw = mk(**kw)
    ref = run(fw, False)
    for rep in range(3):
        got = run(fw, True)
        err = max(float((a[0] - b[0]).abs().max()) for a, b in zip(ref, got))
        gerr = max(float((a[1] - b[1]).abs().max()) for a, b in zip(ref, got))
        nan = any(bool(torch.isnan(b[0]).any()) for b in got)
        print("%-24s rep %d: max|dh| %.3e  max|dgates| %.3e  nan=%s  sync_err=%s" % (name, rep, err, gerr, nan, ops.gru_sync_error()), flush=True)
    Tm = max(d["T"] for d in fw)
    t0 = timeit(lambda: ops.gru_seq_fwd(fw, persistent=False), Tm)
    t1 = timeit(lambda: ops.gru_seq_fwd(fw, persistent=True), Tm)
    print("%-24s per-step %.2f us/step   persistent %.2f us/step" % (name, t0, t1), flush=True)

print("---- backward ----", flush=True)
def mkb(fw, dh0=True, rows=True, ext=True, last=True):
    bw = []
    for d in fw:
        B, T, H = d["B"], d["T"], d["H"]
        w = torch.randn(3*H, H, device=dev) / (H ** 0.5)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        b = dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d.get("h0"), h_all=d["h_all"], gates=d["gates"],
                 dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev))
        if ext: b["dh_ext"] = torch.randn(T, B, H, device=dev) * 0.1
        if last: b["dh_last"] = torch.randn(B, H, device=dev) * 0.1
        if dh0: b["dh0"] = torch.zeros(B, H, device=dev)
        if rows: b["dgx_rowsum"] = torch.zeros(B, 3*H, device=dev); b["dghn_rowsum"] = torch.zeros(B, H, device=dev)
        bw.append(b)
    return bw
OUT = ("dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum")
def runb(bw, persistent):
    for b in bw:
        for k in OUT:
            if b.get(k) is not None: b[k].fill_(0.25 if "rowsum" in k else float("nan"))
        b["scratch"].zero_()
    ops.gru_seq_bwd(bw, persistent=persistent)
    torch.cuda.synchronize()
    return [{k: b[k].clone() for k in OUT if b.get(k) is not None} for b in bw]
for name, kw in cases:
    fw = mk(**kw); run(fw, False)
    for opts in (dict(), dict(dh0=False, rows=False, ext=False)):
        bw = mkb(fw, **opts)
        ref = runb(bw, False)
        for rep in range(2):
            got = runb(bw, True)
            worst = 0.0; nan = False
            for a, b in zip(ref, got):
                for k in a:
                    sc = float(a[k].abs().max()) + 1e-30
                    worst = max(worst, float((a[k] - b[k]).abs().max()) / sc); nan |= bool(torch.isnan(b[k]).any())
            print("%-24s %-10s rep %d: max rel-to-max err %.3e nan=%s sync_err=%s" % (name, "full" if not opts else "minimal", rep, worst, nan, ops.gru_sync_error()), flush=True)
    Tm = max(d["T"] for d in fw)
    t0 = timeit(lambda: ops.gru_seq_bwd(bw, persistent=False), Tm)
    t1 = timeit(lambda: ops.gru_seq_bwd(bw, persistent=True), Tm)
    print("%-24s bwd per-step %.2f us/step   persistent %.2f us/step" % (name, t0, t1), flush=True)
