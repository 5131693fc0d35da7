#!/usr/bin/env python3
"""Generate the golden fixtures that pin the oracle to the reference.

Runs ONLY in the build container (it imports /root/reference); the GPU box never
sees the reference, only the .npz files this script writes next to itself.

What it does
------------
* imports the reference's ``gmm_model.py`` with a ``.cuda()`` no-op shim (the reference
  hard-codes ``.cuda()`` on fresh tensors, gmm_model.py:120,212,214,230);
* AST-extracts ``std_normal, loss_function, latent_regularized_loss_function, train,
  evaluate, convert_to_one_hot`` from ``trainer_gmm.py:101-303`` (the file is a
  run-on-import script, so it cannot be imported) and executes them unmodified in a
  namespace that supplies ``model / optimizer / args``;
* drives them on seeded synthetic batches and dumps inputs, every forward output,
  all loss terms, raw per-parameter gradients, the reference ``train()`` 8-tuples and
  post-Adam weights, and eval-mode greedy decode tokens.

Fixtures
--------
``small.npz``   hidden 64, z 32, B=6, T=20, Tr=8  - weights stored in full.
``c0.npz``      hidden 512, z 128, B=8, T=64, Tr=16 (BASELINE config 0) - weights are
                seeded (torch.manual_seed(1234)); only per-parameter checksums stored.

Usage:  python tests/golden/make_golden.py
"""
import ast
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))

# ---- .cuda() shim (CPU only container) -------------------------------------------------
torch.Tensor.cuda = lambda self, *a, **k: self
torch.nn.Module.cuda = lambda self, *a, **k: self

sys.path.insert(0, REF)
import gmm_model as ref_gmm  # noqa: E402  (the reference model, imported as-is)
from torch import nn, optim  # noqa: E402
from torch.nn import functional as F  # noqa: E402
from torch.distributions import kl_divergence, Normal  # noqa: E402

WANTED = {"std_normal", "loss_function", "latent_regularized_loss_function",
          "train", "evaluate", "convert_to_one_hot"}


def load_trainer_functions(ns):
    """exec the reference's own step/loss functions (trainer_gmm.py:101-303) into ns."""
    src = open(os.path.join(REF, "trainer_gmm.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert {n.name for n in body} == WANTED
    mod = ast.Module(body=body, type_ignores=[])
    exec(compile(mod, "trainer_gmm.py[extract]", "exec"), ns)


def synth_batch(rng, B, T, Tr):
    """SURVEY.md 8(d) synthetic inputs (mirrors ptb_v2.py:261,264,319,352-356,421-422)."""
    d = np.zeros((B, T), np.int64)
    for b in range(B):
        L = int(rng.randint(T // 2, T + 1))
        d[b, :L - 1] = rng.randint(2, 342, size=L - 1)
        d[b, L - 1] = 1
    r = rng.choice(3, size=(B, Tr), p=[.3, .4, .3]).astype(np.int64)
    r[:, 0] = 1
    n = rng.randint(0, 14, size=(B, Tr)).astype(np.int64)
    c = np.zeros((B, 24), np.float32)
    for b in range(B):
        k = int(rng.randint(1, 4))
        pos = rng.choice(24, size=k, replace=False)
        c[b, pos] = rng.uniform(0.1, 1.0, size=k).astype(np.float32)
    r_density = np.array([(row == 1).sum() / Tr for row in r], np.float64)
    n_density = n.mean(axis=1).astype(np.float64)
    a = rng.randint(0, 2, size=(B,)).astype(np.int64)
    return d, r, n, c, r_density, n_density, a


def build(hidden, zdim, K=2, seed=1234):
    torch.manual_seed(seed)
    return ref_gmm.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24,
                                     hidden_dims=hidden, z_dims=zdim, n_step=32, n_component=K)


def checksums(sd):
    out = {}
    for k, v in sd.items():
        v = v.double()
        out[k] = np.array([v.sum().item(), v.abs().sum().item(), (v * v).sum().item()])
    return out


def make_case(name, hidden, zdim, B, T, Tr, store_weights, decode_steps):
    model = build(hidden, zdim)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = {"beta": 0.2, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=nn, F=F, kl_divergence=kl_divergence, Normal=Normal,
              model=model, args=args)
    load_trainer_functions(ns)

    rng = np.random.RandomState(0)
    d, r, n, c, r_den, n_den, a = synth_batch(rng, B, T, Tr)
    td, tr_, tn = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n)
    tc = torch.from_numpy(c)
    ta = torch.from_numpy(a)
    d_oh = ns["convert_to_one_hot"](td, 342)
    r_oh = ns["convert_to_one_hot"](tr_, 3)
    n_oh = ns["convert_to_one_hot"](tn, 16)

    out = {"meta_dims": np.array([hidden, zdim, 2, B, T, Tr]),
           "d": d, "r": r, "n": n, "c": c, "r_density": r_den, "n_density": n_den, "a": a}

    # eps exactly as the reference draws them (gmm_model.py:230,234-235: two randn(B,Z), then
    # T x torch.rand(1) inside the decoder loop, :140)
    torch.manual_seed(99)
    eps_r = torch.randn(B, zdim)
    eps_n = torch.randn(B, zdim)
    out["eps_r"], out["eps_n"] = eps_r.numpy(), eps_n.numpy()

    # ---------------- forward ----------------
    torch.manual_seed(99)
    res = model(d_oh, r_oh, n_oh, tc)
    (o, r_out, n_out, _, _), (dis_r, dis_n), (z_r, z_n), (ll_r, ll_n), (qy_r, qy_n), (y_r, y_n) = res
    # the eps the model drew must be the ones we captured
    assert torch.allclose((z_r - dis_r.mean) / dis_r.stddev, eps_r, atol=1e-4)
    assert torch.allclose((z_n - dis_n.mean) / dis_n.stddev, eps_n, atol=1e-4)
    fw = dict(out=o, r_out=r_out, n_out=n_out, mu_r=dis_r.mean, sigma_r=dis_r.stddev,
              mu_n=dis_n.mean, sigma_n=dis_n.stddev, z_r=z_r, z_n=z_n, ll_r=ll_r, ll_n=ll_n,
              qy_r=qy_r, qy_n=qy_n, y_r=y_r, y_n=y_n)
    for k, v in fw.items():
        out["fw_" + k] = v.detach().numpy()

    # ---------------- losses (reference loss_function / latent_regularized_loss_function) ---
    for step in (0, 5000, 20000):
        ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, (dis_r, dis_n), (qy_r, qy_n),
                                 (ll_r, ll_n), step, beta=args["beta"])
        out["loss_unsup_%d" % step] = np.array([float(x) for x in ls], np.float64)
    ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, (dis_r, dis_n), (qy_r, qy_n),
                             (ll_r, ll_n), 20000, beta=args["beta"], is_supervised=True, y_label=ta)
    out["loss_sup_20000"] = np.array([float(x) for x in ls], np.float64)
    l_r, l_n = ns["latent_regularized_loss_function"]((z_r, z_n), r_den, n_den)
    out["loss_reg"] = np.array([float(l_r), float(l_n)], np.float64)

    # ---------------- raw gradients of the full training loss (unsup + sup, step 20000) -----
    for tag, sup in (("unsup", False), ("sup", True)):
        model.zero_grad()
        torch.manual_seed(99)
        res = model(d_oh, r_oh, n_oh, tc)
        (o, r_out, n_out, _, _), dis, z_out, ll_out, qy_out, _ = res
        ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, dis, qy_out, ll_out, 20000,
                                 beta=args["beta"], is_supervised=sup, y_label=ta if sup else None)
        loss = ls[0]
        l_r, l_n = ns["latent_regularized_loss_function"](z_out, r_den, n_den)
        loss = loss + l_r + l_n
        loss.backward()
        out["total_loss_%s_20000" % tag] = np.array([float(loss)])
        sq = 0.0
        for k, p in model.named_parameters():
            if p.grad is None:
                continue
            g = p.grad.detach()
            sq += float((g.double() ** 2).sum())
            if store_weights:
                out["grad_%s/%s" % (tag, k)] = g.numpy().copy()
            else:
                gd = g.double()
                out["gradsum_%s/%s" % (tag, k)] = np.array([gd.sum().item(), gd.abs().sum().item(),
                                                             (gd * gd).sum().item()])
        out["gradnorm_%s_20000" % tag] = np.array([sq ** 0.5])
        if tag == "unsup":
            out["no_grad_params"] = np.array([k for k, p in model.named_parameters() if p.grad is None])

    # ---------------- the reference's own train(): 3 steps from step 19999, then weights -----
    model.load_state_dict(sd0)
    model.zero_grad()
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    step = 19999
    tuples = []
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = ns["train"](step, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den)
        tuples.append(tup)
    out["train_tuples"] = np.array(tuples, np.float64)
    sd3 = model.state_dict()
    if store_weights:
        for k, v in sd0.items():
            out["w0/" + k] = v.numpy()
        for k, v in sd3.items():
            out["w3/" + k] = v.numpy().copy()
    else:
        for k, v in checksums(sd0).items():
            out["w0sum/" + k] = v
        for k, v in checksums(sd3).items():
            out["w3sum/" + k] = v
    # evaluate() on the trained weights (train mode, step-1 as the reference calls it)
    torch.manual_seed(123)
    out["eval_tuple"] = np.array(ns["evaluate"](step - 1, d_oh, r_oh, n_oh, td, tr_, tn, tc,
                                                r_den, n_den), np.float64)

    # ---------------- eval-mode greedy decode (test_class.py:233-254 path) --------------------
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        dis_r, dis_n = model.encode(d_oh)
        z_r = dis_r.mean + dis_r.stddev * eps_r
        z_n = dis_n.mean + dis_n.stddev * eps_n
        z_r = z_r.clone()
        z_r[:, 0] = 0.75                      # "shifting" (test_class.py:249)
        z = torch.cat([z_r, z_n, tc], dim=1)
        dec = model.global_decoder(z, steps=decode_steps)
    tok = dec.argmax(-1)
    top2 = dec.topk(2, dim=-1).values
    out["dec_z"] = z.numpy()
    out["dec_tokens"] = tok.numpy()
    out["dec_gap"] = (top2[..., 0] - top2[..., 1]).numpy()
    out["dec_logp_first"] = dec[:, 0, :].numpy()
    model.train()

    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(name, "->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
    for k in ("loss_unsup_0", "loss_unsup_5000", "loss_unsup_20000", "loss_sup_20000", "loss_reg",
              "gradnorm_unsup_20000", "train_tuples"):
        print("  ", k, out[k])


def make_hand_cases():
    """convert_to_one_hot (trainer_gmm.py:296) + clean_output (test_class.py:44) known answers."""
    ns = dict(torch=torch, np=np)
    load_trainer_functions_min(ns)
    x = torch.tensor([[0, 3, 1], [2, 2, 0]])
    oh = ns["convert_to_one_hot"](x, 4).numpy()
    v = torch.tensor([1, 0, 3])
    oh1 = ns["convert_to_one_hot"](v, 4).numpy()
    seqs = [[0, 0, 5, 7, 9, 1, 4, 0, 0], [5, 6, 7, 0, 0], [1, 5, 6], [4, 0, 3, 1, 1, 2], [0, 0, 0]]
    cleaned = []
    for s in seqs:
        lp = torch.full((1, len(s), 12), -5.0)
        for i, t in enumerate(s):
            lp[0, i, t] = 0.0
        cleaned.append(ns["clean_output"](lp))
    out = {"oh_in": x.numpy(), "oh_out": oh, "oh1_in": v.numpy(), "oh1_out": oh1}
    for i, (s, cl) in enumerate(zip(seqs, cleaned)):
        out["clean_in_%d" % i] = np.array(s)
        out["clean_out_%d" % i] = np.asarray(cl)
    np.savez_compressed(os.path.join(HERE, "hand.npz"), **out)
    print("hand ->", {k: v.tolist() for k, v in out.items() if k.startswith("clean_out")})


def load_trainer_functions_min(ns):
    src = open(os.path.join(REF, "trainer_gmm.py")).read()
    tree = ast.parse(src)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "convert_to_one_hot"]
    src2 = open(os.path.join(REF, "test_class.py")).read()
    tree2 = ast.parse(src2)
    body += [n for n in tree2.body if isinstance(n, ast.FunctionDef) and n.name == "clean_output"]
    exec(compile(ast.Module(body=body, type_ignores=[]), "extract", "exec"), ns)


if __name__ == "__main__":
    torch.set_num_threads(8)
    make_hand_cases()
    make_case("small", hidden=64, zdim=32, B=6, T=20, Tr=8, store_weights=True, decode_steps=30)
    make_case("c0", hidden=512, zdim=128, B=8, T=64, Tr=16, store_weights=False, decode_steps=100)
