"""Randomised stress of the weight-stationary scans against the per-step kernels: random scan counts / batch rows / lengths / H,
random input kinds, forward + backward, repeated launches on the same buffers (L1/L2-warm exchange slabs), optional CU budgets."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
V = 57
rng = np.random.RandomState(int(os.environ.get("SEED", "0")))
budget_s = float(os.environ.get("FUZZ_SECONDS", "120"))
t_end = time.time() + budget_s
n_cases = n_bad = 0
worst = 0.0
while time.time() < t_end:
    H = int(rng.choice([64, 96, 128, 256, 512]))
    n = int(rng.randint(1, 5))
    B = int(rng.choice([1, 5, 16, 33, 64, 100, 128, 200, 256]))
    budget = int(rng.choice([0, 0, 128, 64]))
    scans, bws = [], []
    for s in range(n):
        T = int(rng.randint(2, 40))
        w = (torch.randn(3*H, H, device=dev) / (H ** 0.5)).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=B, T=T, H=H, reverse=int(rng.randint(2)), w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1,
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        if rng.rand() < 0.7: d["b_ih"] = torch.randn(3*H, device=dev) * 0.1
        if rng.rand() < 0.6: d["h0"] = torch.randn(B, H, device=dev) * 0.3
        kind = rng.randint(3)
        if kind in (0, 2):
            d["gx_table"] = torch.randn(V, 3*H, device=dev) * 0.3
            d["idx"] = torch.randint(0, V, (B, T + 3), dtype=torch.int32, device=dev)
            if rng.rand() < 0.3: d["idx_shift"], d["start_token"] = -1, V - 1
        if kind in (1, 2): d["gx_dense"] = torch.randn(T, B, 3*H, device=dev) * 0.3
        if rng.rand() < 0.5: d["gx_rowbias"] = torch.randn(B, 3*H, device=dev) * 0.2
        scans.append(d)
        b = dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d.get("h0"), h_all=d["h_all"], gates=d["gates"],
                 dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev))
        if rng.rand() < 0.7: b["dh_ext"] = torch.randn(T, B, H, device=dev) * 0.1
        if rng.rand() < 0.7: b["dh_last"] = torch.randn(B, H, device=dev) * 0.1
        if rng.rand() < 0.6: b["dh0"] = torch.zeros(B, H, device=dev)
        if rng.rand() < 0.6: b["dgx_rowsum"] = torch.zeros(B, 3*H, device=dev); b["dghn_rowsum"] = torch.zeros(B, H, device=dev)
        bws.append(b)
    def run(persistent):
        out = []
        for rep in range(3 if persistent else 1):              # repeated launches: warm caches, reused slabs
            for d in scans: d["h_all"].fill_(float("nan"))
            ops.gru_seq_fwd(scans, persistent=persistent, cu_budget=budget)
            for b in bws:
                for k in ("dgx_all", "dghn_all", "dh0"):
                    if b.get(k) is not None: b[k].fill_(float("nan"))
                for k in ("dgx_rowsum", "dghn_rowsum"):
                    if b.get(k) is not None: b[k].zero_()
                b["scratch"].zero_()
            ops.gru_seq_bwd(bws, persistent=persistent, cu_budget=budget)
            torch.cuda.synchronize()
            out.append([d["h_all"].clone() for d in scans] + [d["gates"].clone() for d in scans] +
                       [b[k].clone() for b in bws for k in ("dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum") if b.get(k) is not None])
        return out
    ref = run(False)[0]
    for got in run(True):
        for a, b in zip(ref, got):
            sc = float(a.abs().max()) + 1e-20
            e = float((a - b).abs().max()) / sc if not torch.isnan(b).any() else float("inf")
            worst = max(worst, e) if e != float("inf") else worst
            if not (e < 5e-5):
                n_bad += 1
                print("MISMATCH H=%d n=%d B=%d budget=%d err=%g Ts=%s" % (H, n, B, budget, e, [d["T"] for d in scans]), flush=True)
                break
    n_cases += 1
print("fuzz: %d random cases x 3 repeated launches, %d mismatches, worst rel-to-max error %.2e, sync_err=%s" % (n_cases, n_bad, worst, ops.gru_sync_error()))
