"""A/B of where the decoder-side weight-gradient GEMMs run (Engine.dw_order x Engine.lean_dw) on the whole training step (C1 shape).
usage: python scratch/ab_dw.py side:1 before:0 after:0 ...   (order:lean[:splitk override])"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
for rep in range(2):
  for spec in sys.argv[1:] or ["side:1"]:
    parts = spec.split(":")
    order, lean = parts[0], int(parts[1])
    torch.manual_seed(1234)
    m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    eng = m.engine()
    eng.dw_order, eng.lean_dw = order, bool(lean)
    for kv in parts[2:]:
        k, v = kv.split("=")
        setattr(eng, k, type(getattr(eng, k))(int(v)))
    b = synth_batch(np.random.RandomState(0), 256, 256, 64)
    batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
    torch.manual_seed(99); eps = tr.draw_eps(256, 256)
    step = 20000
    for _ in range(4):
        tr.step_device(step, batch, eps); step += 1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        tr.step_device(step, batch, eps); step += 1
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print("%-24s %.3f ms/step  loss %.4f" % (spec, dt * 1e3, tr._tuple8(0.2, 256, False)[0]), flush=True)
    del tr, m, eng
