#!/usr/bin/env python3
"""Golden fixture for the BENCHMARK configuration (BASELINE configs[1]): hidden 512, z 128, K=2, B=256, T=256, Tr=64.

Runs ONLY in the build container (imports /root/reference through make_golden.py's shim + AST extraction) and takes a few
minutes of CPU time (one reference step at this size is ~1 minute on 8 cores).  Only checksums are stored (``c1.npz`` is a few
hundred KB: the batch itself is re-derived from the seeds by the test / bench.py, but kept in the file so the fixture is
self-contained).

What is pinned, with the seeds ``bench.py`` uses (weights ``torch.manual_seed(1234)``, data ``RandomState(0)``, eps
``torch.manual_seed(99)`` before EVERY step, i.e. the same eps every step, first training step = 20000):
  * raw gradients of one forward/backward of the full training loss: total norm + per-parameter (sum, |sum|, sum of squares);
  * the reference's own ``train()`` (trainer_gmm.py:220-258) called twice: both 8-tuples, post-step weight checksums;
  * forward checksums of that first call's outputs (out / r_out / n_out sums, z, logLogit, q(y|x), y).
"""
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the .cuda() shim and imports the reference model)
from torch import nn, optim  # noqa: E402
from torch.nn import functional as F  # noqa: E402
from torch.distributions import kl_divergence, Normal  # noqa: E402


def csum(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()])


def main(hidden=512, zdim=128, B=256, T=256, Tr=64, name="c1"):
    t0 = time.time()
    model = mg.build(hidden, zdim)
    model.train()
    args = {"beta": 0.2, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=nn, F=F, kl_divergence=kl_divergence, Normal=Normal, model=model, args=args)
    mg.load_trainer_functions(ns)
    d, r, n, c, r_den, n_den, a = mg.synth_batch(np.random.RandomState(0), B, T, Tr)
    td, tr_, tn, tc = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)
    d_oh, r_oh, n_oh = (ns["convert_to_one_hot"](x, v) for x, v in ((td, 342), (tr_, 3), (tn, 16)))
    out = {"meta_dims": np.array([hidden, zdim, 2, B, T, Tr]), "d": d.astype(np.int16), "r": r.astype(np.int8), "n": n.astype(np.int8),
           "c": c, "r_density": r_den, "n_density": n_den}
    for k, v in mg.checksums(model.state_dict()).items():
        out["w0sum/" + k] = v

    # ---- one forward/backward of the full training loss at step 20000 (raw, unclipped gradients) ----
    torch.manual_seed(99)
    res = model(d_oh, r_oh, n_oh, tc)
    (o, r_out, n_out, _, _), dis, z_out, ll_out, qy_out, y_out = res
    for k, v in dict(out=o, r_out=r_out, n_out=n_out, mu_r=dis[0].mean, sigma_r=dis[0].stddev, mu_n=dis[1].mean, sigma_n=dis[1].stddev,
                     z_r=z_out[0], z_n=z_out[1], ll_r=ll_out[0], ll_n=ll_out[1], qy_r=qy_out[0], qy_n=qy_out[1]).items():
        out["fwsum_" + k] = csum(v)
    out["fw_y_r"], out["fw_y_n"] = y_out[0].numpy().astype(np.int8), y_out[1].numpy().astype(np.int8)
    out["fw_out_row0"] = o[0, :4].detach().numpy()              # a few full log-prob rows
    ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, dis, qy_out, ll_out, 20000, beta=args["beta"])
    l_r, l_n = ns["latent_regularized_loss_function"](z_out, r_den, n_den)
    loss = ls[0] + l_r + l_n
    loss.backward()
    out["total_loss_20000"] = np.array([float(loss)])
    sq = 0.0
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        out["gradsum/" + k] = csum(p.grad)
        sq += float((p.grad.double() ** 2).sum())
    out["gradnorm_20000"] = np.array([sq ** 0.5])
    print("fwd/bwd done %.0f s: loss %.6f gradnorm %.6f" % (time.time() - t0, float(loss), sq ** 0.5), flush=True)

    # ---- the reference's own train(), twice, same eps every step (what bench.py does) ----
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    step, tuples = 20000, []
    for it in range(2):
        torch.manual_seed(99)
        step, tup = ns["train"](step, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den)
        tuples.append(tup)
        print("train step %d done %.0f s: %s" % (it, time.time() - t0, tup), flush=True)
        for k, v in mg.checksums(model.state_dict()).items():
            out["w%dsum/%s" % (it + 1, k)] = v
    out["train_tuples"] = np.array(tuples, np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
