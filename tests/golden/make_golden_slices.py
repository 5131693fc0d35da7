#!/usr/bin/env python3
"""Element-wise gradient slices at hidden 512 (``slices.npz``): c0 / c1 store per-parameter CHECKSUMS only, which a row permutation
or a sign pattern inside one tensor would pass.  This fixture adds, from the reference itself (build container only, make_golden.py's shim):

``c0u/``, ``c0s/``  BASELINE configs[0] shape (hidden 512, z 128, B=8, T=64, Tr=16), unsupervised / supervised loss at step 20000
``c1u/``            the BENCHMARK shape (B=256, T=256, Tr=64), unsupervised loss at step 20000
    ``grad/<param>``          first 2 + last 2 rows (or the whole tensor when it has < 8 rows / is 1-D and short) ...
    ``gstride/<param>``       ... and every 97th element of the flattened gradient
    ``xgrad_r``, ``xgrad_n``  d loss / d (input of gru_r / gru_n) at the (batch, time) positions ``xg_b`` x ``xg_t``: both directions'
                              gate-gradient rows projected through W_ih (= dgx_fwd[t, b] W_ih + dgx_bwd[t, b] W_ih_reverse), taken with a
                              forward pre-hook that hands the encoder a fresh leaf copy of x (the arithmetic stays the reference's)
    ``loss``, ``gradnorm``
``fader/``          MusicAttrFaderNets (model_v2.py:438) at hidden 512, z 128, B=256, T=64: trainer_fader.py's own loss at step 20000:
                    the same slices, the loss terms, ONE reference ``train()`` tuple - the 128-row tiles of the single-encoder engine.
"""
import ast
import os
import sys
import time
from collections import Counter

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from torch import optim  # noqa: E402

STRIDE = 97


def put_slices(out, pfx, named_grads):
    for k, g in named_grads:
        g = g.detach().numpy()
        if g.ndim == 2 and g.shape[0] >= 8:
            out[pfx + "grad/" + k] = np.concatenate([g[:2], g[-2:]], 0).copy()
        elif g.size <= 4096:
            out[pfx + "grad/" + k] = g.copy()
        out[pfx + "gstride/" + k] = g.reshape(-1)[::STRIDE].copy()


def tap_input(mod, store, key):
    def hook(m, inp):
        leaf = inp[0].detach().clone().requires_grad_(True)
        store[key] = leaf
        return (leaf,) + tuple(inp[1:])
    return mod.register_forward_pre_hook(hook)


def gmvae_case(out, pfx, hidden, zdim, B, T, Tr, sup):
    t0 = time.time()
    model = mg.build(hidden, zdim)
    model.train()
    args = {"beta": 0.2, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, model=model, args=args)
    mg.load_trainer_functions(ns)
    d, r, n, c, r_den, n_den, a = mg.synth_batch(np.random.RandomState(0), B, T, Tr)
    td, tr_, tn, tc, ta = (torch.from_numpy(x) for x in (d, r, n, c, a))
    d_oh, r_oh, n_oh = (ns["convert_to_one_hot"](x, v) for x, v in ((td, 342), (tr_, 3), (tn, 16)))
    leaves = {}
    hooks = [tap_input(model.gru_r, leaves, "r"), tap_input(model.gru_n, leaves, "n")]
    torch.manual_seed(99)
    res = model(d_oh, r_oh, n_oh, tc)
    (o, r_out, n_out, _, _), dis, z_out, ll_out, qy_out, _ = res
    ls = ns["loss_function"](o, td, r_out, tr_, n_out, tn, dis, qy_out, ll_out, 20000, beta=args["beta"],
                             is_supervised=sup, y_label=ta if sup else None)
    l_r, l_n = ns["latent_regularized_loss_function"](z_out, r_den, n_den)
    loss = ls[0] + l_r + l_n
    loss.backward()
    for h in hooks:
        h.remove()
    out[pfx + "dims"] = np.array([hidden, zdim, 2, B, T, Tr])
    out[pfx + "loss"] = np.array([float(loss)])
    grads = [(k, p.grad) for k, p in model.named_parameters() if p.grad is not None]
    out[pfx + "gradnorm"] = np.array([sum(float((g.double() ** 2).sum()) for _, g in grads) ** 0.5])
    put_slices(out, pfx, grads)
    bs = sorted({0, 1, B // 2 - 1, B // 2, B - 2, B - 1})
    ts = sorted({0, 1, T // 3, 2 * T // 3, T - 2, T - 1})
    out[pfx + "xg_b"], out[pfx + "xg_t"] = np.array(bs), np.array(ts)
    for key in ("r", "n"):
        out[pfx + "xgrad_" + key] = leaves[key].grad[bs][:, ts].numpy().copy()       # [len(bs)][len(ts)][342]
    print("%s done %.0f s: loss %.6f gradnorm %.6f" % (pfx, time.time() - t0, float(loss), float(out[pfx + "gradnorm"][0])), flush=True)


def fader_case(out, pfx, hidden=512, zdim=128, B=256, T=64, Tr=16):
    import model_v2 as ref_v2
    t0 = time.time()
    torch.manual_seed(1234)
    model = ref_v2.MusicAttrFaderNets(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=hidden, z_dims=zdim, n_step=T)
    model.train()
    args = {"beta": 0.2, "lr": 1e-3}
    wanted = {"std_normal", "loss_function", "adversarial_loss", "train", "evaluate"}
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, Counter=Counter, model=model, args=args, step=0)
    body = [n for n in ast.parse(open(os.path.join(mg.REF, "trainer_fader.py")).read()).body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted
    exec(compile(ast.Module(body=body, type_ignores=[]), "trainer_fader.py[extract]", "exec"), ns)
    d, r, n, c, r_den, n_den, _ = mg.synth_batch(np.random.RandomState(5), B, T, Tr)
    td, tr_, tn, tc = (torch.from_numpy(x) for x in (d, r, n, c))
    oh = lambda x, dims: torch.zeros(tuple(x.shape) + (dims,)).scatter_(-1, x.unsqueeze(-1), 1.0)
    d_oh, r_oh, n_oh = oh(td, 342), oh(tr_, 3), oh(tn, 16)
    rd_arg, nd_arg = torch.from_numpy(r_den).float().unsqueeze(-1), torch.from_numpy(n_den).float().unsqueeze(-1)      # trainer_fader.py:180
    out[pfx + "dims"] = np.array([hidden, zdim, B, T, Tr])
    for k, v in mg.checksums(model.state_dict()).items():
        out[pfx + "w0sum/" + k] = v
    torch.manual_seed(99)
    (o, r_out, n_out), dis, z = model(d_oh, r_oh, n_oh, tc, rd_arg, nd_arg)
    loss, ce = ns["loss_function"](o, td, dis, 20000, beta=args["beta"])
    la_r, la_n = ns["adversarial_loss"](20000, r_out, n_out, rd_arg, nd_arg)
    total = loss + la_r + la_n
    total.backward()
    out[pfx + "loss_terms"] = np.array([float(x) for x in (total, ce, la_r, la_n)])
    grads = [(k, p.grad) for k, p in model.named_parameters() if p.grad is not None]
    out[pfx + "gradnorm"] = np.array([sum(float((g.double() ** 2).sum()) for _, g in grads) ** 0.5])
    put_slices(out, pfx, grads)
    out[pfx + "fw_z"], out[pfx + "fw_mu"] = z.detach().numpy(), dis.mean.detach().numpy()
    out[pfx + "fw_r_out"] = r_out.detach().numpy()
    out[pfx + "fw_out_rows"] = o[[0, B // 2, B - 1]][:, [0, T // 2, T - 1]].detach().numpy()
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    torch.manual_seed(99)
    step, tup = ns["train"](19999, d_oh, r_oh, n_oh, td, tr_, tn, tc, rd_arg, nd_arg)
    out[pfx + "train_tuple"] = np.array(tup, np.float64)
    for k, v in mg.checksums(model.state_dict()).items():
        out[pfx + "w1sum/" + k] = v
    print("%s done %.0f s: loss terms %s train %s" % (pfx, time.time() - t0, out[pfx + "loss_terms"], tup), flush=True)


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    gmvae_case(out, "c0u/", 512, 128, 8, 64, 16, sup=False)
    gmvae_case(out, "c0s/", 512, 128, 8, 64, 16, sup=True)
    fader_case(out, "fader/")
    gmvae_case(out, "c1u/", 512, 128, 256, 256, 64, sup=False)
    path = os.path.join(HERE, "slices.npz")
    np.savez_compressed(path, **out)
    print("slices ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
