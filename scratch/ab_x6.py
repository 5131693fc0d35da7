"""round 4: the opt-in bf16 x 6 weight-gradient products (HipOps.dw_x6) - isolated dW_hh-shaped product and the whole training step, alternating in one session"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from mfn_import import load_package
pkg = load_package()
from music_fader_nets_amd.hipops import HipOps
from music_fader_nets_amd.synth import synth_batch
dev = torch.device("cuda:0"); ops = HipOps(dev)
H, rows = 512, 65280
torch.manual_seed(5)
dgx, dghn, hp = torch.randn(rows, 3 * H, device=dev), torch.randn(rows, H, device=dev), torch.randn(rows, H, device=dev) * 0.3
dW = torch.zeros(3 * H, H, device=dev)
want = torch.cat([dgx[:, :2 * H], dghn], 1).double().t() @ hp.double()
sc = float((torch.cat([dgx[:, :2 * H], dghn], 1).double().abs().t() @ hp.double().abs()).max())
for x6 in (False, True, False, True):
    ops.dw_x6 = x6
    for _ in range(3): ops.gru_dwhh(dgx, dghn, hp, dW, splitk=16)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gru_dwhh(dgx, dghn, hp, dW, splitk=16)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    err = float((dW.double() - want).abs().max()) / sc
    print("dW_hh product 1536 x 512 x %d, bf16x6=%d: %.1f us per launch = %.1f fp32-equivalent TFLOP/s;  max error vs float64 / max sum|a||b| = %.3e" % (rows, x6, us, 2.0 * 3 * H * H * rows / us / 1e6, err), flush=True)
del dgx, dghn, hp
res = {}
for rep in range(3):
    for x6 in (False, True):
        torch.manual_seed(1234)
        m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, 512, 128, 32, n_component=2).to(dev)
        tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
        m.engine().ops.dw_x6 = x6
        m.weights_changed()
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
        print("training step, bf16x6 weight gradients=%d: %.3f ms/step  loss after 34 steps %.6f" % (x6, dt * 1e3, tr._tuple8(0.2, 256, False)[0]), flush=True)
        res.setdefault(x6, []).append(dt * 1e3)
        del tr, m
print({k: ["%.3f" % x for x in v] for k, v in res.items()})
