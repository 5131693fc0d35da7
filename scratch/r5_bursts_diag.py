"""round 5: which gradient goes nondeterministic in the eager data-parallel step of tests/test_parallel_rccl.py::test_bursts_... (one failure in
31 soak runs)?  One rank through RCCL, eager launches, the test's shape; REPS fresh trainers run 8 steps each from the same weights and inputs,
after every step the flat GRADIENT is compared with the first trainer's gradient of that step; differing parameters are named.
  python scratch/r5_bursts_diag.py [arith] [reps] [mode]      mode: none | joined
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
arith = sys.argv[1] if len(sys.argv) > 1 else "bf16x6"
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 60
MODE = sys.argv[3] if len(sys.argv) > 3 else "none"
SNAP = len(sys.argv) > 4 and sys.argv[4] == "snap"
NSTEP = int(os.environ.get("NSTEP", "8"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", FN_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                  FN_DP_GRAPH="0")
from helpers import make_model  # noqa: E402
from mfn_import import load_package  # noqa: E402
pkg = load_package()
if os.environ.get("FN_LIB"):                     # A/B of library builds (scratch/r5_build_variant*.sh)
    from music_fader_nets_amd import _lib
    _lib.LIB_PATH = os.path.join(ROOT, "scratch", os.environ["FN_LIB"])
    print("library:", _lib.LIB_PATH, flush=True)
from music_fader_nets_amd import parallel  # noqa: E402
from music_fader_nets_amd.synth import synth_batch  # noqa: E402

if MODE in ("graph", "eager"):
    ctx, local = None, 0
else:
    ctx, local = parallel.init_from_env()
dev = "cuda:%d" % local
B, T, Tr, NB, CYC = 256, 64, 16, 9, 600_000
b = synth_batch(np.random.RandomState(0), B, T, Tr)
first = None
bad = 0
for rep in range(REPS):
    m = make_model(512, 128, device=dev, arith=arith)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=ctx)
    if MODE == "eager":
        tr.use_graph = False
    ops = m.engine().ops
    if os.environ.get("DIAG_SERIALIZE") == "1":
        m.engine().serialize_lanes = True
    if os.environ.get("DIAG_VARIANT"):
        ops.variant = int(os.environ["DIAG_VARIANT"], 0)
    if os.environ.get("DIAG_NOSAFE") == "1":       # the register-stationary backward in eager launches too (what Engine.eager_safe_bwd avoids)
        m.engine().eager_safe_bwd = False
    if os.environ.get("DIAG_NOFILL") == "1":
        m.engine().fill_edges = False
    plain_finish = type(ctx).finish_buckets if ctx is not None else None
    if MODE == "joined":
        def finish(self, ops=ops):
            with torch.cuda.stream(self.rccl.stream):
                for _ in range(NB):
                    ops.occupy_cus(8, 64 * 1024, CYC)
            plain_finish(self)
        ctx.finish_buckets = finish.__get__(ctx)
    batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
    torch.manual_seed(99)
    eps = tr.draw_eps(B, T)
    grads = []
    snaps = []
    eng = m.engine()
    for s in range(NSTEP):
        tr.step_device(20000 + s, batch, eps)
        grads.append(tr.flat.grad.clone())
        if SNAP:                                  # every named scratch buffer of the engine after this step (stream-ordered clones on the main stream)
            eng.main_wait_side()
            snaps.append({k: v.clone() for k, v in eng._bufs.items() if v.is_floating_point()})
    torch.cuda.synchronize()
    if SNAP and first is None:
        first_snaps = snaps
    assert not ops.gru_sync_error()
    if MODE == "joined":
        del ctx.finish_buckets
    if first is None:
        first = grads
        continue
    for s in range(NSTEP):
        if not torch.equal(grads[s], first[s]):
            bad += 1
            names = []
            for k in tr.flat.names:
                g0 = tr.flat.G[k]
                off = g0.data_ptr() - tr.flat.grad.data_ptr()
                off //= 4
                a, c = grads[s][off:off + g0.numel()], first[s][off:off + g0.numel()]
                if not torch.equal(a, c):
                    d = (a - c).abs()
                    names.append("%s: %d of %d elements, max |diff| %.3e (max |g| %.3e), first at %d" % (k, int((d > 0).sum()), d.numel(), float(d.max()), float(c.abs().max()),
                                                                                                    int(torch.nonzero(d > 0)[0])))
            print("rep %d step %d DIFFERS:\n  %s" % (rep, s, "\n  ".join(names)), flush=True)
            if SNAP:
                for k, v in snaps[s].items():
                    w = first_snaps[s].get(k)
                    if w is None or w.shape != v.shape:
                        continue
                    neq = (v != w) & ~(torch.isnan(v) & torch.isnan(w))
                    if bool(neq.any()):
                        idx = torch.nonzero(neq)
                        lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
                        print("    buffer %-28s %s: %d elements differ, index box %s .. %s, max |diff| %.3e (max |ref| %.3e)" % (
                            k[0], tuple(v.shape), int(neq.sum()), lo, hi, float((v - w)[neq].abs().max()), float(w.abs().max())), flush=True)
                        if v.dim() == 3 and k[0] in ("g_dgx1", "sd_dgx_r", "g_dgx2", "sd_dgx_n", "g_dghn1", "sd_dghn_r"):
                            # the first step that went wrong (time runs backwards): which rows and which columns differ there, and how much
                            for t in range(hi[0], max(hi[0] - 3, -1), -1):
                                m2 = neq[t]
                                cols = torch.nonzero(m2.any(0)).view(-1).tolist()
                                rows = torch.nonzero(m2.any(1)).view(-1).tolist()
                                runs, start = [], None
                                for c in cols + [None]:
                                    if start is None:
                                        start = prev = c
                                    elif c is not None and c == prev + 1:
                                        prev = c
                                    else:
                                        runs.append("%d-%d" % (start, prev))
                                        start = prev = c
                                rel = float(((v[t] - w[t]).abs().max()) / w[t].abs().max())
                                print("        t = %d: rows %d..%d (%d), %d columns %s, max |diff| / max |ref| of the step %.3e" % (
                                    t, rows[0], rows[-1], len(rows), len(cols), " ".join(runs[:40]), rel), flush=True)
            break
print("%s, mode %s: %d of %d repetitions differ from the first" % (arith, MODE, bad, REPS - 1), flush=True)
if ctx is not None:
    ctx.rccl.close()
    torch.distributed.destroy_process_group()
