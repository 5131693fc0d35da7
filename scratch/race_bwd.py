"""which backward scan form is the flaky one?  test_ping_pong_scans...[4-256-Ts0-kinds0]'s backward, repeated: every variant against the round-3 loop (0x800)"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
from mfn_import import load_package
load_package()
import test_gpu_parity as tp
from music_fader_nets_amd.hipops import HipOps
DEV = torch.device("cuda:0"); ops = HipOps(DEV)
H, n, B = 512, 4, 256
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
scans = tp._pp_forward_scans(ops, n, B, H, (9,), 7 * n + B, ("table", "table_rev"))
ops.gru_seq_fwd(scans, variant=0x800)
torch.manual_seed(3 * n + B)
bw = []
for i, d in enumerate(scans):
    T = d["T"]
    wt = torch.zeros(ops.frag_floats(H, 3 * H), device=DEV)
    ops.frag_pack((torch.randn(H, 3 * H, device=DEV) / H ** 0.5).contiguous(), wt)
    bw.append(dict(B=B, T=T, H=H, w_hh_t_frag=wt, h0=d.get("h0"), h_all=d["h_all"], gates=d["gates"],
                   dh_last=torch.randn(B, H, device=DEV) if i % 3 != 1 else None, dh_ext=torch.randn(T, B, H, device=DEV) * 0.1 if i % 3 != 0 else None,
                   dgx_all=torch.zeros(T, B, 3 * H, device=DEV), dghn_all=torch.zeros(T, B, H, device=DEV),
                   dh0=torch.zeros(B, H, device=DEV) if (d.get("h0") is not None or i == 0) else None,
                   dgx_rowsum=torch.zeros(B, 3 * H, device=DEV) if i & 1 else None, dghn_rowsum=torch.zeros(B, H, device=DEV) if i != 2 else None,
                   scratch=torch.zeros(B, H, device=DEV)))
outs = ("dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum")
def clear():
    for b in bw:
        for k in outs:
            if b[k] is not None:
                b[k].zero_() if "rowsum" in k else b[k].fill_(float("nan"))
clear(); ops.gru_seq_bwd(bw, variant=0x800)
refb = [{k: b[k].clone() for k in outs if b[k] is not None} for b in bw]
for var in (0x800, 0x2000):
    bad = 0
    for rep in range(reps):
        clear()
        ops.gru_seq_bwd(bw, variant=var)
        ok = all(torch.equal(b[k], v) for r, b in zip(refb, bw) for k, v in r.items())
        bad += (not ok)
    print("variant %#06x: %d of %d launches differ from the round-3 loop   sync error %s" % (var, bad, reps, ops.gru_sync_error()), flush=True)

# the default (register-stationary) backward: bit-stable from launch to launch?
clear(); ops.gru_seq_bwd(bw)
first = [{k: b[k].clone() for k in outs if b[k] is not None} for b in bw]
bad = 0
for rep in range(reps):
    clear(); ops.gru_seq_bwd(bw)
    bad += not all(torch.equal(b[k], v) for r, b in zip(first, bw) for k, v in r.items())
print("default backward (gru_bwd_rs_kernel): %d of %d launches differ from the first   sync error %s" % (bad, reps, ops.gru_sync_error()), flush=True)
# the default forward (gru_fwd_pp_kernel) against the round-3 loop, and the opt-in x6 forward against itself
ref = [(d["h_all"].clone(), d["gates"].clone()) for d in scans]
for name, x6 in (("gru_fwd_pp_kernel", False), ("gru_fwd_x6_kernel", True)):
    ops.dw_x6 = x6
    if x6:
        for d in scans:
            d["w_hh_frag3"] = torch.zeros(ops.frag_floats(3 * H, H) * 3 // 2, device=DEV)
        # triple images from the fp32 fragments are not available here: pack from a fresh matrix for both
        for d in scans:
            w = (torch.randn(3 * H, H, device=DEV) / 22).contiguous()
            ops.frag_pack(w, d["w_hh_frag"]); ops.frag3_pack(w, d["w_hh_frag3"])
        ops.gru_seq_fwd(scans)
        ref = [(d["h_all"].clone(), d["gates"].clone()) for d in scans]
    bad = 0
    for rep in range(reps):
        for d in scans:
            d["h_all"].fill_(float("nan")); d["gates"].fill_(float("nan"))
        ops.gru_seq_fwd(scans)
        bad += not all(torch.equal(d["h_all"], h) and torch.equal(d["gates"], gt) for (h, gt), d in zip(ref, scans))
    print("forward %s: %d of %d launches differ   sync error %s" % (name, bad, reps, ops.gru_sync_error()), flush=True)
ops.dw_x6 = False
