"""round 5: the decoder backward pipeline of the engine (its real lanes, launches and GEMMs) repeated on ONE saved forward state, eager launches, T = 64 /
Tr = 16 (the shape of tests/test_parallel_rccl.py::test_bursts_...): which piece makes one 16-row tile of one step go wrong once in ~2500 steps?
Every repetition's gate gradients are compared bit for bit with the first one's.
  python scratch/r5_bwd_pipeline_loop.py [batches of 20] [knob ...]     knobs: noaxpy  nogemm1 (the aux GEMM of launch k >= 1 runs on the main stream)
                                                                              serial (all lanes on the main stream)  x6  variant=0x...
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import make_model  # noqa: E402
from mfn_import import load_package  # noqa: E402
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch  # noqa: E402

BATCHES = int(sys.argv[1]) if len(sys.argv) > 1 else 100
KNOBS = set(sys.argv[2:])
NB = 20
dev = "cuda:0"
B, T, Tr = 256, 64, 16
for k in list(KNOBS):
    if k.startswith("T="):
        T = int(k[2:])
    if k.startswith("Tr="):
        Tr = int(k[3:])
b = synth_batch(np.random.RandomState(0), B, T, Tr)
m = make_model(512, 128, device=dev, arith="bf16x6" if "x6" in KNOBS else "f32")
tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
tr.use_graph = False
eng = m.engine()
ops = eng.ops
for k in KNOBS:
    if k.startswith("variant="):
        ops.variant = int(k[8:], 0)
if "serial" in KNOBS:
    eng.serialize_lanes = True
batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
torch.manual_seed(99)
eps = tr.draw_eps(B, T)
tr.loss_and_grads(20000, batch, eps)             # buffers exist, weights packed
torch.cuda.synchronize()
fw = tr._forward_losses(20000, batch, eps, want_grads=True)
eng.main_wait_side()
torch.cuda.synchronize()
S = eng.saved
sd = S["dec"]["sd"]

if "noaxpy" in KNOBS:
    ops.axpy = lambda *a, **k: None
if "nogemm1" in KNOBS:
    real_lane = pkg.engine.Engine._Lane if hasattr(pkg, "engine") else None
    import music_fader_nets_amd.engine as E_
    orig_enter = E_.Engine._Lane.__enter__

    def enter(self):
        if self.lane.startswith("auxb"):
            self.side = False
        return orig_enter(self)
    E_.Engine._Lane.__enter__ = enter

ref = None
bad = 0
names = ("dgx1", "dgx2", "sd_r", "sd_n")
for bt in range(BATCHES):
    outs = []
    for i in range(NB):
        if "hostbound" in KNOBS:          # the GPU waits for the host as in a real eager step: every kernel starts when it is enqueued
            torch.cuda.synchronize()
        eng.main_wait_side()
        sdb, sds = eng._bwd_sub_decoder_scans(sd, fw[0], B, Tr, defer=True)
        gd = eng._bwd_global_decoder_scans(S, fill={e: (sds[e], sdb[e]["dh0"]) for e in ("r", "n")})
        outs.append([gd["dgx1"].clone(), gd["dgx2"].clone(), sdb["r"]["dgx"].clone(), sdb["n"]["dgx"].clone()])
    torch.cuda.synchronize()
    if ref is None:
        ref = outs[0]
    for i, o in enumerate(outs):
        for nm, a, r in zip(names, o, ref):
            if not torch.equal(a, r):
                bad += 1
                neq = a != r
                idx = torch.nonzero(neq)
                hi = idx.max(0).values.tolist()
                m2 = neq[hi[0]]
                cols = torch.nonzero(m2.any(0)).view(-1)
                rows = torch.nonzero(m2.any(1)).view(-1)
                print("batch %d rep %d %s: first wrong t = %d, rows %d..%d, %d columns from %d, max |diff| / max |ref| there %.2e" % (
                    bt, i, nm, hi[0], int(rows[0]), int(rows[-1]), cols.numel(), int(cols[0]), float((a[hi[0]] - r[hi[0]]).abs().max() / r[hi[0]].abs().max())), flush=True)
                break
assert not ops.gru_sync_error()
print("knobs %s: %d of %d repetitions differ from the first" % (sorted(KNOBS), bad, BATCHES * NB), flush=True)
