"""round 5: does the hand-over of the weight-stationary scans depend on WHERE the workgroups of a row group sit?  (one 16-row tile of one step went
wrong once in ~1000 eager data-parallel steps: scratch/r5_bursts_diag.py.)  The same launch repeated REPS times with the default placement (a row
group on one XCD) and with variant bit 12 (every row group spread over all XCDs), alone and beside a GEMM on another stream; every result is
compared bit for bit with the first one.
  python scratch/r5_xcd_spread.py [reps]
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
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ops = HipOps(torch.device(DEV))
H = 512


def make(n, B, T, seed):
    torch.manual_seed(seed)
    fwd, bwd = [], []
    for s_ in range(n):
        w = (torch.randn(3 * H, H, device=DEV) / (H ** 0.5)).contiguous()
        wf = torch.zeros(ops.frag_floats(3 * H, H), device=DEV)
        ops.frag_pack(w, wf)
        wf3 = torch.zeros(ops.frag_floats(3 * H, H) * 3 // 2, device=DEV)
        wt = torch.zeros(ops.frag_floats(H, 3 * H), device=DEV)
        wt3 = torch.zeros(ops.frag_floats(H, 3 * H) * 3 // 2, device=DEV)
        ops.weight_images([("frag_t", w, wt), ("frag3_t", w, wt3), ("frag3", w, wf3)])
        h0 = torch.randn(B, H, device=DEV) * 0.3
        f = dict(B=B, T=T, H=H, w_hh_frag=wf, w_hh_frag3=wf3, b_hh=torch.randn(3 * H, device=DEV) * 0.1, b_ih=torch.randn(3 * H, device=DEV) * 0.1, h0=h0,
                 gx_dense=torch.randn(T, B, 3 * H, device=DEV) * 0.5, h_all=torch.zeros(T, B, H, device=DEV), gates=torch.zeros(T, ops.gates_floats(B, H), device=DEV))
        fwd.append(f)
        bwd.append(dict(B=B, T=T, H=H, w_hh_t_frag=wt, w_hh_t_frag3=wt3, h0=h0, h_all=f["h_all"], gates=f["gates"],
                        dh_last=torch.randn(B, H, device=DEV), dh_ext=torch.randn(T, B, H, device=DEV) * 0.5,
                        dgx_all=torch.zeros(T, B, 3 * H, device=DEV), dghn_all=torch.zeros(T, B, H, device=DEV),
                        dh0=torch.zeros(B, H, device=DEV), dgx_rowsum=torch.zeros(B, 3 * H, device=DEV),
                        dghn_rowsum=torch.zeros(B, H, device=DEV), scratch=torch.zeros(B, H, device=DEV)))
    return fwd, bwd


side = torch.cuda.Stream(device=DEV)
A = torch.randn(8192, 1536, device=DEV)
Wt = torch.randn(512, 1536, device=DEV)
C = torch.zeros(8192, 512, device=DEV)


def beside(on):
    if not on:
        return
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.lane = "aux/"
        ops.gemm(A, Wt, C, a_k=True, b_k=True)
        ops.lane = ""


def run_case(name, n, B, T, x6, which):
    fwd, bwd = make(n, B, T, 7)
    ops.dw_x6, ops.variant = x6, 0
    ops.gru_seq_fwd(fwd)
    keys_b = ("dgx_all", "dghn_all", "dh0", "dgx_rowsum", "dghn_rowsum")
    ref = None
    for variant in (0, 0x1000):
        for co in (False, True):
            bad = 0
            where = None
            for rep in range(REPS):
                ops.variant = variant
                if which == "bwd":
                    for b in bwd:
                        b["dgx_rowsum"].zero_()
                        b["dghn_rowsum"].zero_()
                    beside(co)
                    ops.gru_seq_bwd(bwd)
                    out = [b[k].clone() for b in bwd for k in keys_b]
                else:
                    beside(co)
                    ops.gru_seq_fwd(fwd)
                    out = [f[k].clone() for f in fwd for k in ("h_all", "gates")]
                torch.cuda.current_stream().wait_stream(side)
                if ref is None:
                    ref = out
                    continue
                for i, (a, b) in enumerate(zip(out, ref)):
                    if not torch.equal(a, b):
                        bad += 1
                        if where is None:
                            d = torch.nonzero(a != b)
                            where = (i, d.min(0).values.tolist(), d.max(0).values.tolist(), float((a - b).abs().max()), float(b.abs().max()))
                        break
            torch.cuda.synchronize()
            assert not ops.gru_sync_error()
            print("%-34s %s  placement %-6s  %-12s: %d of %d launches differ from the first%s" % (
                name, "bf16x6" if x6 else "f32   ", "spread" if variant else "xcd", "beside a GEMM" if co else "alone", bad, REPS,
                "" if where is None else "  (first: output %d, index box %s .. %s, max |diff| %.3e of %.3e)" % where), flush=True)
    ops.variant = 0


for x6 in (False, True):
    run_case("decoder backward 2 x 256 x 32", 2, 256, 32, x6, "bwd")
    run_case("encoder backward 4 x 256 x 64", 4, 256, 64, x6, "bwd")
    run_case("decoder forward 2 x 256 x 32", 2, 256, 32, x6, "fwd")
    run_case("encoder forward 4 x 256 x 64", 4, 256, 64, x6, "fwd")
