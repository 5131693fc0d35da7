"""round 4: isolated timing of the weight-stationary scans, ping-pong form (variant 0) against the round-3 loops (variant 0x800):
encoder shape (4 scans x 256 rows, T = 256) and decoder shape (2 scans x 256 rows, T = 32), forward and backward; us per launch and
fraction of the fp32 MFMA peak.  usage: python scratch/pp_time.py [fwd|bwd|both]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, V = 512, 342
which = sys.argv[1] if len(sys.argv) > 1 else "both"

def mk(n, B, T):
    fw, bw = [], []
    for s in range(n):
        w = (torch.randn(3*H, H, device=dev) / 22).contiguous()
        wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
        wtf = torch.zeros(ops.frag_floats(H, 3*H), device=dev); ops.frag_pack(w.t().contiguous(), wtf)
        d = dict(B=B, T=T, H=H, reverse=s & 1, w_hh_frag=wf, b_hh=torch.zeros(3*H, device=dev), b_ih=torch.zeros(3*H, device=dev),
                 gx_table=torch.randn(V, 3*H, device=dev) * 0.1, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                 h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
        if n == 2:
            d["h0"] = torch.randn(B, H, device=dev) * 0.1
        fw.append(d)
        bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wtf, h0=d.get("h0"), h_all=d["h_all"], gates=d["gates"], dh_ext=torch.randn(T, B, H, device=dev) * 0.01,
                       dgx_all=torch.zeros(T, B, 3*H, device=dev), dghn_all=torch.zeros(T, B, H, device=dev), scratch=torch.zeros(B, H, device=dev),
                       dh0=torch.zeros(B, H, device=dev) if n == 2 else None,
                       dgx_rowsum=torch.zeros(B, 3*H, device=dev), dghn_rowsum=torch.zeros(B, H, device=dev)))
    return fw, bw

def timeit(fn, reps):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps

for name, n, B, T, reps in (("encoder 4 x 256 rows, T=256", 4, 256, 256, 10), ("decoder 2 x 256 rows, T=32", 2, 256, 32, 40)):
    fw, bw = mk(n, B, T)
    flop = n * B * T * 2.0 * H * 3 * H
    for direction, fn_of in (("fwd", lambda v: (lambda: ops.gru_seq_fwd(fw, variant=v))), ("bwd", lambda v: (lambda: ops.gru_seq_bwd(bw, variant=v)))):
        if which not in ("both", direction):
            continue
        if direction == "bwd":
            ops.gru_seq_fwd(fw, variant=0x800)
        res = {}
        for rnd in range(2):
            for v, tag in ((0x800, "round-3 loop"), (0, "ping-pong")):
                res.setdefault(tag, []).append(timeit(fn_of(v), reps))
        for tag, us in res.items():
            print("%-30s %s %-13s %s us per launch -> %s of the fp32 MFMA peak" % (name, direction, tag, " / ".join("%.1f" % u for u in us),
                  " / ".join("%.3f" % (flop / (u * 1e-6) / 157.3e12) for u in us)))
assert not ops.gru_sync_error()
