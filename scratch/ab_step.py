"""A/B of a scan tiling variant on the whole training step (C1 shape): ms/step with ops.variant = 0 and the given values.
usage: python scratch/ab_step.py 0 256 ...   (values of FnGruFwd.variant; bit 8 = the round-1 64-row tiling <4,1,1>)"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import shutil
libs = [a for a in sys.argv[1:] if a.endswith(".so")]
if libs:          # A/B of builds inside one gpurun call: python ab_step.py scratch/lib_x.so [variants...]
    shutil.copy(os.path.join(R, libs[0]), os.path.join(R, "music-fader-nets_amd/libfadernets_hip.so"))
    sys.argv = [a for a in sys.argv if not a.endswith(".so")]
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0")
for variant in [int(x) for x in sys.argv[1:]] or [0]:
    for los in (True,):
        torch.manual_seed(1234)
        m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
        tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
        m.engine().ops.variant = variant
        m.engine().fused_head = los
        b = synth_batch(np.random.RandomState(0), 256, 256, 64)
        batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
        torch.manual_seed(99); eps = tr.draw_eps(256, 256)
        step = 20000
        for _ in range(3):
            tr.step_device(step, batch, eps); step += 1
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            tr.step_device(step, batch, eps); step += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print("variant %d fused_head %d: %.3f ms/step  loss %.4f" % (variant, los, dt * 1e3, tr._tuple8(0.2, 256, False)[0]), flush=True)
        del tr, m
