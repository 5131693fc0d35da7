#!/usr/bin/env python3
"""Golden fixture for the GLSR trainer (reference ``trainer_glsr.py`` on ``model_v2.MusicAttrRegVAE``): ``glsr.npz``.

Runs ONLY in the build container.  ``loss_function / latent_regularized_loss_function / train / evaluate / std_normal`` are AST-extracted
from ``trainer_glsr.py`` (a run-on-import script) and executed unmodified.

The reference's rhythm-density walk (:139-165) only does something when the decoder puts >= 0.9 of its probability on the time-shift
tokens at some steps; a randomly initialised output layer never does (0.29), and the regulariser would be a constant.  The fixture therefore
scales the output layer's weight of the seeded model (``out_scale``) and shifts its bias (time-shift tokens 180..277 up, note-on tokens 2..89
down; stored as ``bias_shift``) so that every branch of the walk is taken: steps below and above the 0.9 threshold, flushes with accumulated note mass above and below 1e-2.
Hidden 64, z 32, B=4, T=104 (>= the 100 teacher-forced steps of the GLSR decodes), Tr=8.
"""
import ast
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import model_v2 as ref_v2  # noqa: E402
from torch import optim  # noqa: E402

H, Z, B, T, TR = 64, 32, 4, 104, 8
OUT_SCALE, SEP_SHIFT, NOTE_SHIFT = 16.0, 2.8, -1.7
WANTED = {"std_normal", "loss_function", "latent_regularized_loss_function", "train", "evaluate"}


def main():
    torch.manual_seed(1234)
    model = ref_v2.MusicAttrRegVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T)
    shift = torch.zeros(342)
    shift[180:278] = SEP_SHIFT
    shift[2:90] = NOTE_SHIFT
    with torch.no_grad():
        model.linear_out_g.weight.mul_(OUT_SCALE)
        model.linear_out_g.bias.add_(shift)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = {"beta": 0.2, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, model=model, args=args)
    body = [n for n in ast.parse(open(os.path.join(mg.REF, "trainer_glsr.py")).read()).body if isinstance(n, ast.FunctionDef) and n.name in WANTED]
    assert {n.name for n in body} == WANTED
    exec(compile(ast.Module(body=body, type_ignores=[]), "trainer_glsr.py[extract]", "exec"), ns)

    d, r, n, c, r_den, n_den, _ = mg.synth_batch(np.random.RandomState(11), B, T, TR)
    td, tr_, tn, tc = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)
    oh = lambda x, dims: torch.zeros(tuple(x.shape) + (dims,)).scatter_(-1, x.unsqueeze(-1), 1.0)
    d_oh, r_oh, n_oh = oh(td, 342), oh(tr_, 3), oh(tn, 16)
    out = {"dims": np.array([H, Z, B, T, TR]), "d": d, "r": r, "n": n, "c": c, "r_density": r_den, "n_density": n_den,
           "bias_shift": shift.numpy(), "out_scale": np.array([OUT_SCALE])}
    for k, v in mg.checksums(sd0).items():
        out["w0sum/" + k] = v

    # ---- one forward + loss + GLSR at step 20000, raw gradients -------------------------------------------------------------------
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(99)
    (o, r_out, n_out), dis, z_out = model(d_oh, r_oh, n_oh, tc)
    loss, ce_x, ce_r, ce_n = ns["loss_function"](o, td, r_out, tr_, n_out, tn, dis, 20000, beta=args["beta"])
    l_r, l_n = ns["latent_regularized_loss_function"](z_out, r_den, n_den, tc)
    total = loss + l_r + l_n
    total.backward()
    out["loss_terms_20000"] = np.array([float(total), float(ce_x), float(ce_r), float(ce_n), float(l_r), float(l_n)])
    sq = 0.0
    for k, p in model.named_parameters():
        if p.grad is not None:
            out["grad/" + k] = p.grad.numpy().copy()
            sq += float((p.grad.double() ** 2).sum())
    out["gradnorm"] = np.array([sq ** 0.5])
    # how much of the walk the fixture exercises (diagnostic, also asserted by the tests)
    with torch.no_grad():
        torch.manual_seed(5)
        zc = torch.cat([z_out[0], z_out[1], tc], dim=1)
        probs = model.global_decoder(zc, steps=100).exp()
        sep = probs[:, :, 180:278].sum(-1)
        out["diag_sep_frac_above"] = np.array([float((sep >= 0.9).float().mean())])
        out["diag_note_mass"] = np.array([float(probs[:, :, 2:90].sum(-1).mean())])
    print("loss terms", out["loss_terms_20000"], "gradnorm", out["gradnorm"], "sep>=0.9 share", out["diag_sep_frac_above"], "mean note mass", out["diag_note_mass"])

    # ---- the reference's own train(): step 19 (regulariser off, :289-292) and 3 steps from 19999, then evaluate() -------------------------
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    torch.manual_seed(50)
    _, tup = ns["train"](19, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den)
    out["train_tuple_step19"] = np.array(tup, np.float64)
    model.load_state_dict(sd0)
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    step, tuples = 19999, []
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = ns["train"](step, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den)
        tuples.append(tup)
    out["train_tuples"] = np.array(tuples, np.float64)
    for k, v in mg.checksums(model.state_dict()).items():
        out["w3sum/" + k] = v
    torch.manual_seed(123)
    with torch.no_grad():
        out["eval_tuple"] = np.array(ns["evaluate"](step - 1, d_oh, r_oh, n_oh, td, tr_, tn, tc, r_den, n_den), np.float64)
    print("train tuples", np.array(tuples), "\neval", out["eval_tuple"], "\nstep19", out["train_tuple_step19"])
    path = os.path.join(HERE, "glsr.npz")
    np.savez_compressed(path, **out)
    print("glsr ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
