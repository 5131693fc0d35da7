#!/usr/bin/env python3
"""Golden fixture for the vanilla-VAE sibling: imports the reference's ``model_v2.MusicAttrRegVAE`` and runs the
reference's own ``loss_function / latent_regularized_loss_function / train / evaluate`` of ``trainer.py`` (AST-extracted, executed
unmodified, module-level ``step = 0`` as in trainer.py:57) on a seeded synthetic batch.  Output: vae.npz next to this file.

Runs ONLY in the build container.   python tests/golden/make_golden_vae.py
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (.cuda() shim, sys.path to the reference, synth_batch)
import model_v2 as ref_v2  # noqa: E402

H, Z, B, T, TR = 64, 32, 6, 20, 8
WANTED = {"std_normal", "loss_function", "latent_regularized_loss_function", "train", "evaluate"}


def main():
    torch.manual_seed(1234)
    model = ref_v2.MusicAttrRegVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = {"beta": 0.1, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, model=model, args=args, step=0)
    src = open(os.path.join(mg.REF, "trainer.py")).read()
    body = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert {n.name for n in body} == WANTED
    exec(compile(ast.Module(body=body, type_ignores=[]), "trainer.py[extract]", "exec"), ns)

    rng = np.random.RandomState(3)
    d, r, n, c, r_den, n_den, _a = mg.synth_batch(rng, B, T, TR)
    td, tr_, tn, tc = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)
    oh = lambda x, dims: torch.zeros(tuple(x.shape) + (dims,)).scatter_(-1, x.unsqueeze(-1), 1.0)
    d_oh, r_oh, n_oh = oh(td, 342), oh(tr_, 3), oh(tn, 16)
    out = {"meta_dims": np.array([H, Z, 1, B, T, TR]), "d": d, "r": r, "n": n, "c": c, "r_density": r_den, "n_density": n_den}

    torch.manual_seed(99)
    eps_r, eps_n = torch.randn(B, Z), torch.randn(B, Z)
    out["eps_r"], out["eps_n"] = eps_r.numpy(), eps_n.numpy()
    torch.manual_seed(99)
    (o, r_out, n_out), (dis_r, dis_n), (z_r, z_n) = model(d_oh, r_oh, n_oh, tc)
    # Normal(0,1).sample(size) (model_v2.py:152-154) must consume the generator exactly like randn(size)
    assert torch.allclose((z_r - dis_r.mean) / dis_r.stddev, eps_r, atol=1e-4)
    assert torch.allclose((z_n - dis_n.mean) / dis_n.stddev, eps_n, atol=1e-4)
    fw = dict(out=o, r_out=r_out, n_out=n_out, mu_r=dis_r.mean, sigma_r=dis_r.stddev, mu_n=dis_n.mean, sigma_n=dis_n.stddev, z_r=z_r, z_n=z_n)
    for k, v in fw.items():
        out["fw_" + k] = v.detach().numpy()
    # loss terms + raw gradients of the reference's own step (before clipping)
    for p in model.parameters():
        p.grad = None
    ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, (dis_r, dis_n), beta=args["beta"])
    l_r, l_n = ns["latent_regularized_loss_function"]((z_r, z_n), r_den, n_den)
    total = ls[0] + l_r + l_n
    out["loss_terms"] = np.array([float(total), float(ls[1]), float(ls[2]), float(ls[3]), float(l_r), float(l_n)])
    total.backward()
    gn = 0.0
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = p.grad.numpy().copy()
            gn += float((p.grad.double() ** 2).sum())
    out["gradnorm"] = np.array([gn ** 0.5])
    # three optimisation steps + evaluate with the reference's train()/evaluate()
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = mg.optim.Adam(model.parameters(), lr=args["lr"])
    step, tuples = 5000, []
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = ns["train"](step, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den)
        tuples.append(tup)
    out["train_tuples"] = np.array(tuples, np.float64)
    torch.manual_seed(123)
    out["eval_tuple"] = np.array(ns["evaluate"](d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den), np.float64)
    for k, v in sd0.items():                      # seeded construction: checksums only (sum, sum of |.|)
        out["w0sum/" + k] = np.array([float(v.double().sum()), float(v.double().abs().sum())])
    for k, v in model.state_dict().items():
        out["w3/" + k] = v.numpy().copy()
    path = os.path.join(HERE, "vae.npz")
    np.savez_compressed(path, **out)
    print("vae ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
    print("loss terms", out["loss_terms"], "gradnorm", out["gradnorm"], "\ntrain", out["train_tuples"], "\neval", out["eval_tuple"])
    print("params without grad:", [k for k, p in model.named_parameters() if "grad/" + k not in out])


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
