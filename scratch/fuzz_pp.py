"""round 4: randomised stress of the ping-pong / register-stationary weight-stationary scans (gru_fwd_pp_kernel<1|2>, gru_bwd_rs_kernel<1|2>, gru_bwd_pp_kernel) at H = 512
against the per-step kernels: random scan counts, batch rows that make full 64- / 128-row groups, lengths, input kinds, optional h0 / row bias / shifted tokens, three repeated
launches on the same buffers (warm exchange slabs), forward also bit for bit against the round-3 loops (variant 0x400).  FUZZ_SECONDS, SEED from the environment."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 57
rng = np.random.RandomState(int(os.environ.get("SEED", "0")))
t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", "120"))
n_cases = n_bad = n_bit = 0
worst = 0.0
shapes = {}
while time.time() < t_end:
    B = int(rng.choice([64, 128, 192, 256]))
    nmax = {64: 8, 128: 8, 192: 5, 256: 4}[B]
    n = int(rng.randint(1, nmax + 1))
    bvar = int(rng.choice([0, 0, 0x2000]))                    # backward: default dispatch (rs where eligible) or the 32-slice loop
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
        if rng.rand() < 0.5:
            d["gx_table"] = torch.randn(V, 3*H, device=dev) * 0.3
            d["idx"] = torch.randint(0, V, (B, T + 3), dtype=torch.int32, device=dev)
            if rng.rand() < 0.3 and not d["reverse"]: d["idx_shift"], d["start_token"] = -1, V - 1
        else:
            d["gx_dense"] = torch.randn(T, B, 3*H, device=dev) * 0.3
        if rng.rand() < 0.5: d["gx_rowbias"] = torch.randn(B, 3*H, device=dev) * 0.2
        scans.append(d)
        b = dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d.get("h0"), h_all=d["h_all"], gates=d["gates"],
                 dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev))
        if rng.rand() < 0.7: b["dh_ext"] = torch.randn(T, B, H, device=dev) * 0.1
        if rng.rand() < 0.7: b["dh_last"] = torch.randn(B, H, device=dev) * 0.1
        if rng.rand() < 0.6: b["dh0"] = torch.zeros(B, H, device=dev)
        if rng.rand() < 0.6: b["dgx_rowsum"] = torch.zeros(B, 3*H, device=dev); b["dghn_rowsum"] = torch.zeros(B, H, device=dev)
        bws.append(b)
    shapes[(n, B)] = shapes.get((n, B), 0) + 1

    def run(persistent, fvar=0, reps=1):
        out = []
        for rep in range(reps):
            for d in scans: d["h_all"].fill_(float("nan"))
            ops.gru_seq_fwd(scans, persistent=persistent, variant=fvar)
            for b in bws:
                for k in ("dgx_all", "dghn_all", "dh0"):
                    if b.get(k) is not None: b[k].fill_(float("nan"))
                for k in ("dgx_rowsum", "dghn_rowsum"):
                    if b.get(k) is not None: b[k].zero_()
                b["scratch"].zero_()
            ops.gru_seq_bwd(bws, persistent=persistent, variant=bvar)
            torch.cuda.synchronize()
            out.append(([d["h_all"].clone() for d in scans] + [d["gates"].clone() for d in scans],
                        [b[k].clone() for b in bws for k in ("dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum") if b.get(k) is not None]))
        return out
    ref = run(False)[0]
    old = run(True, fvar=0x400)[0]
    for got in run(True, reps=3):
        if all(torch.equal(a, b) for a, b in zip(old[0], got[0])):
            n_bit += 1
        else:
            n_bad += 1
            print("forward NOT bit-identical to the round-3 loop: n=%d B=%d Ts=%s" % (n, B, [d["T"] for d in scans]), flush=True)
        for a, b in zip(ref[0] + ref[1], got[0] + got[1]):
            sc = float(a.abs().max()) + 1e-20
            e = float((a - b).abs().max()) / sc if not torch.isnan(b).any() else float("inf")
            worst = max(worst, e) if e != float("inf") else worst
            if not (e < 5e-5):
                n_bad += 1
                print("MISMATCH n=%d B=%d bvar=%#x err=%g Ts=%s" % (n, B, bvar, e, [d["T"] for d in scans]), flush=True)
                break
    n_cases += 1
print("fuzz_pp: %d random cases x 3 repeated launches at H = 512 (scans x rows: %s), %d mismatches, forward bit-identical to the round-3 loops in %d of %d launches, "
      "worst rel-to-max error vs the per-step kernels %.2e, sync_err=%s" % (n_cases, sorted(shapes.items()), n_bad, n_bit, 3 * n_cases, worst, ops.gru_sync_error()))
