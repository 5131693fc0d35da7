"""round 5: the failing configuration of scratch/r5_bursts_diag.py in isolation - the decoder backward's launch k = 2 at T = 64 / Tr = 16: layer 1's
chunk (256 rows x 32 steps) + the rhythm decoder (256 rows x 16 steps) in ONE gru_bwd_rs_kernel<1> launch, beside the projection GEMM of the aux
lane (issued first, as in the step); batches of 40 back-to-back launches (no host sync inside a batch), every result compared bit for bit with
the first.
  python scratch/r5_xcd_stress2.py [batches] [variant] [gemm: 1 | 0]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from mfn_import import load_package  # noqa: E402
load_package()
from music_fader_nets_amd.hipops import HipOps  # noqa: E402

DEV = "cuda:0"
BATCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 100
VARIANT = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
GEMM = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
PRE = (sys.argv[4] != "0") if len(sys.argv) > 4 else False        # a one-scan launch (128 workgroups: the older 32-slice kernel) in front of every launch, as launch k = 1 of the step
NB = 40
ops = HipOps(torch.device(DEV))
ops.dw_x6 = False
ops.variant = VARIANT
H, B = 512, 256
torch.manual_seed(3)
fwd, bwd = [], []
for T in (32, 16, 32):
    w = (torch.randn(3 * H, H, device=DEV) / (H ** 0.5)).contiguous()
    wf = torch.zeros(ops.frag_floats(3 * H, H), device=DEV)
    ops.frag_pack(w, wf)
    wt = torch.zeros(ops.frag_floats(H, 3 * H), device=DEV)
    ops.weight_images([("frag_t", w, wt)])
    h0 = torch.randn(B, H, device=DEV) * 0.3
    f = dict(B=B, T=T, H=H, w_hh_frag=wf, b_hh=torch.randn(3 * H, device=DEV) * 0.1, b_ih=torch.randn(3 * H, device=DEV) * 0.1, h0=h0,
             gx_dense=torch.randn(T, B, 3 * H, device=DEV) * 0.5, h_all=torch.zeros(T, B, H, device=DEV), gates=torch.zeros(T, ops.gates_floats(B, H), device=DEV))
    fwd.append(f)
    bwd.append(dict(B=B, T=T, H=H, w_hh_t_frag=wt, h0=h0, h_all=f["h_all"], gates=f["gates"], dh_last=None, dh_ext=torch.randn(T, B, H, device=DEV) * 0.5,
                    dgx_all=torch.zeros(T, B, 3 * H, device=DEV), dghn_all=torch.zeros(T, B, H, device=DEV), dh0=torch.zeros(B, H, device=DEV),
                    dgx_rowsum=torch.zeros(B, 3 * H, device=DEV), dghn_rowsum=torch.zeros(B, H, device=DEV), scratch=torch.zeros(B, H, device=DEV)))
ops.gru_seq_fwd(fwd)
side = torch.cuda.Stream(device=DEV)
A = torch.randn(8192, 1536, device=DEV)
Wt = torch.randn(512, 1536, device=DEV)
C = torch.zeros(8192, 512, device=DEV)
C2 = torch.zeros(8192, 512, device=DEV)
main = torch.cuda.current_stream()
ref = None
bad = 0
for bt in range(BATCHES):
    outs = []
    for i in range(NB):
        for b in bwd:
            b["dgx_rowsum"].zero_()
            b["dghn_rowsum"].zero_()
        if GEMM:
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.lane = "aux/"
                ops.gemm(A, Wt, C, a_k=True, b_k=True)
                ops.lane = ""
        if PRE:
            ops.gru_seq_bwd(bwd[2:])
            if GEMM:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ops.lane = "aux/"
                    ops.gemm(A, Wt, C2, a_k=True, b_k=True)
                    ops.lane = ""
        ops.gru_seq_bwd(bwd[:2])
        outs.append([b["dgx_all"].clone() for b in bwd[:2]])
        main.wait_stream(side)
    torch.cuda.synchronize()
    if ref is None:
        ref = outs[0]
    for i, o in enumerate(outs):
        for s_, (a, r) in enumerate(zip(o, ref)):
            if not torch.equal(a, r):
                bad += 1
                d = torch.nonzero(a != r)
                print("batch %d launch %d scan %d: index box %s .. %s, max |diff| %.3e of %.3e" % (bt, i, s_, d.min(0).values.tolist(), d.max(0).values.tolist(),
                                                                                                 float((a - r).abs().max()), float(r.abs().max())), flush=True)
assert not ops.gru_sync_error()
print("variant 0x%x, %s, %s: %d of %d launches differ from the first" % (VARIANT, "beside the GEMM" if GEMM else "alone", "behind a one-scan launch" if PRE else "no launch in front", bad, BATCHES * NB), flush=True)
