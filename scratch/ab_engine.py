"""A/B of Engine schedule switches on the whole training step (C1 shape), alternating inside ONE gpurun call (box-to-box spread is larger than most effects).
usage: python scratch/ab_engine.py "" "losses_early=False" "dw_order='side_late',losses_early=False" ...   (comma-separated attr=python-literal; "" = defaults)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
if os.environ.get("FN_LIB"):
    from music_fader_nets_amd import _lib
    _lib.LIB_PATH = os.path.join(R, "scratch", os.environ["FN_LIB"]); print("library:", _lib.LIB_PATH)
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
REPS = int(os.environ.get("AB_REPS", "3"))
res = {}
for rep in range(REPS):
    for spec in sys.argv[1:] or [""]:
        torch.manual_seed(1234)
        m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
        tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
        if os.environ.get("AB_ARITH"):
            m.set_arith(os.environ["AB_ARITH"])
        eng = m.engine()
        for kv in [x for x in spec.split(",") if x]:
            k, v = kv.split("=", 1)
            if k.startswith("ops."):
                setattr(eng.ops, k[4:], eval(v))
            else:
                setattr(eng, k, eval(v))
        b = synth_batch(np.random.RandomState(0), 256, 256, 64)
        batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
        torch.manual_seed(99); eps = tr.draw_eps(256, 256)
        step = 20000
        for _ in range(4):
            tr.step_device(step, batch, eps); step += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(30):
            tr.step_device(step, batch, eps); step += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
        loss = tr._tuple8(0.2, 256, False)[0]
        res.setdefault(spec, []).append(dt * 1e3)
        print("%-60s %.3f ms/step  loss %.6f" % (spec or "(defaults)", dt * 1e3, loss), flush=True)
        del tr, m, eng
print()
for spec, v in res.items():
    print("%-60s %s   min %.3f" % (spec or "(defaults)", " ".join("%.3f" % x for x in v), min(v)))
