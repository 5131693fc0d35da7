"""Two weight-stationary scans at once on two streams (half the CUs each) + a GEMM on a third: results vs the same scans run alone."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
load_package()
from music_fader_nets_amd.hipops import HipOps
dev = torch.device("cuda:0"); ops = HipOps(dev)
V, H = 57, 512
rng = np.random.RandomState(3)
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
def mk(B, T):
    w = (torch.randn(3*H, H, device=dev) / (H ** 0.5)).contiguous()
    wf = torch.zeros(ops.frag_floats(3*H, H), device=dev); ops.frag_pack(w, wf)
    return dict(B=B, T=T, H=H, reverse=0, w_hh_frag=wf, b_hh=torch.randn(3*H, device=dev) * 0.1, h0=torch.randn(B, H, device=dev) * 0.3,
                gx_table=torch.randn(V, 3*H, device=dev) * 0.3, idx=torch.randint(0, V, (B, T), dtype=torch.int32, device=dev),
                h_all=torch.zeros(T, B, H, device=dev), gates=torch.zeros(T, ops.gates_floats(B, H), device=dev))
A = torch.randn(8192, 1536, device=dev); Bm = torch.randn(8192, 512, device=dev); Cm = torch.zeros(1536, 512, device=dev)
bad = 0; n = 0; t_end = time.time() + float(os.environ.get("FUZZ_SECONDS", "60"))
while time.time() < t_end:
    B = int(rng.choice([64, 128, 256])); T1, T2 = int(rng.randint(8, 70)), int(rng.randint(8, 70))
    a, b = mk(B, T1), mk(B, T2)
    ops.lane = ""; ops.gru_seq_fwd([a], cu_budget=128); ops.gru_seq_fwd([b], cu_budget=128); torch.cuda.synchronize()
    ra, rb = a["h_all"].clone(), b["h_all"].clone()
    a["h_all"].fill_(float("nan")); b["h_all"].fill_(float("nan"))
    cur = torch.cuda.current_stream()
    for s in (s1, s2, s3): s.wait_stream(cur)
    with torch.cuda.stream(s1):
        ops.lane = "l1/"; ops.gru_seq_fwd([a], cu_budget=128)
    with torch.cuda.stream(s3):
        ops.lane = "g/"; ops.gemm(A, Bm, Cm, a_k=False, b_k=False, splitk=4)
    with torch.cuda.stream(s2):
        ops.lane = "l2/"; ops.gru_seq_fwd([b], cu_budget=128)
    for s in (s1, s2, s3): cur.wait_stream(s)
    torch.cuda.synchronize(); ops.lane = ""
    if not (torch.equal(ra, a["h_all"]) and torch.equal(rb, b["h_all"])):
        bad += 1; print("MISMATCH B=%d T=%d,%d" % (B, T1, T2), float((ra - a["h_all"]).abs().max()), float((rb - b["h_all"]).abs().max()), flush=True)
    n += 1
print("concurrent: %d cases, %d mismatches (bit-exact comparison), sync_err=%s" % (n, bad, ops.gru_sync_error()))
