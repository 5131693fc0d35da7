import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from helpers import *
from fake_ops import FakeOps
pkg = load_package()
gold = load_golden("small")
def run(dev, ops=None):
    m = make_model(64, 32, sd_from(gold, "w0/"), device=dev, ops=ops)
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    b = batch_of(gold)
    batch = tr.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"], None)
    eps = (torch.from_numpy(gold["eps_r"]).to(dev), torch.from_numpy(gold["eps_n"]).to(dev))
    tr.loss_and_grads(20000, batch, eps)
    return {k: tr.flat.G[k].cpu().numpy().copy() for k in tr.flat.names}, m
gg, m = run("cuda:0")
gf, _ = run("cpu", FakeOps())
for k in gg:
    ref = gold["grad_unsup/" + k]
    print("%-34s max|ref| %.3e  gpu-vs-ref %.2e  fake-vs-ref %.2e  gpu-vs-fake %.2e" % (k, np.abs(ref).max(), relerr(gg[k], ref), relerr(gf[k], ref), relerr(gg[k], gf[k])))
k = "gru_n.weight_ih_l0"
ref = gold["grad_unsup/" + k]; d = np.abs(gg[k] - ref)
i = np.unravel_index(d.argmax(), d.shape); print("worst", i, gg[k][i], gf[k][i], ref[i], "token col count", (gold["d"] == i[1]).sum())
# compare the intermediate dgx of the encoder between gpu and fake
