#!/usr/bin/env python3
"""Golden fixtures for the single-encoder siblings: ``siblings.npz``.

Runs ONLY in the build container.  Imports the reference's ``model_v2`` (``.cuda()`` shim of make_golden.py) and executes the
reference's own ``loss_function / latent_regularized_loss_function / adversarial_loss / train / evaluate`` AST-extracted from
``trainer_singlevae.py``, ``trainer_cvae.py`` and ``trainer_fader.py`` (run-on-import scripts), unmodified, on a seeded synthetic batch.

Per model ("single", "cvae", "fader"; hidden 64, z 32, B=6, T=20, Tr=8), prefix ``<tag>/``:
  seeded initial weights (checksums), forward outputs of a seeded call, the raw gradients of the training loss at step 20000
  (Fader: also at step 500, where the adversarial weight is still on its ramp), three ``train()`` calls (tuples + checksums of the final weights),
  one ``evaluate()`` call, an eval-mode forward (greedy decoder).
"""
import ast
import os
import sys
from collections import Counter

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import model_v2 as ref_v2  # noqa: E402
from torch import optim  # noqa: E402

H, Z, B, T, TR = 64, 32, 6, 20, 8
CASES = {
    "single": ("MusicAttrSingleVAE", "trainer_singlevae.py", {"std_normal", "loss_function", "latent_regularized_loss_function", "train", "evaluate"}),
    "cvae": ("MusicAttrCVAE", "trainer_cvae.py", {"std_normal", "loss_function", "train", "evaluate"}),
    "fader": ("MusicAttrFaderNets", "trainer_fader.py", {"std_normal", "loss_function", "adversarial_loss", "train", "evaluate"}),
}


def make(tag, out):
    cls, script, wanted = CASES[tag]
    torch.manual_seed(1234)
    model = getattr(ref_v2, cls)(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = {"beta": 0.2, "lr": 1e-3}
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, Counter=Counter, model=model,
              args=args, step=0)
    body = [n for n in ast.parse(open(os.path.join(mg.REF, script)).read()).body if isinstance(n, ast.FunctionDef) and n.name in wanted]
    assert {n.name for n in body} == wanted, (tag, {n.name for n in body})
    exec(compile(ast.Module(body=body, type_ignores=[]), script + "[extract]", "exec"), ns)

    d, r, n, c, r_den, n_den, _ = mg.synth_batch(np.random.RandomState(5), B, T, TR)
    td, tr_, tn, tc = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)
    oh = lambda x, dims: torch.zeros(tuple(x.shape) + (dims,)).scatter_(-1, x.unsqueeze(-1), 1.0)
    d_oh, r_oh, n_oh = oh(td, 342), oh(tr_, 3), oh(tn, 16)
    if tag == "single":
        rd_arg, nd_arg = r_den, n_den                                       # float64 numpy, as the loader yields (trainer_singlevae.py:107-120)
        call = lambda: model(d_oh, tc)
    else:
        rd_arg, nd_arg = torch.from_numpy(r_den).float().unsqueeze(-1), torch.from_numpy(n_den).float().unsqueeze(-1)   # trainer_cvae.py:199-200
        call = lambda: model(d_oh, r_oh, n_oh, tc, rd_arg, nd_arg)
    P = tag + "/"
    out[P + "dims"] = np.array([H, Z, B, T, TR])
    for k, v in (("d", d), ("r", r), ("n", n), ("c", c), ("r_density", r_den), ("n_density", n_den)):
        out[P + k] = v
    for k, v in mg.checksums(sd0).items():
        out[P + "w0sum/" + k] = v

    def total_loss(step):
        res = call()
        if tag == "single":
            o, dis, z = res
            loss, ce = ns["loss_function"](o, td, dis, step, beta=args["beta"])
            l_r, l_n = ns["latent_regularized_loss_function"](z, r_den, n_den)
            return loss + l_r + l_n, (loss + l_r + l_n, ce, l_r, l_n), res
        if tag == "cvae":
            o, dis, z = res
            loss, ce = ns["loss_function"](o, td, dis, step, beta=args["beta"])
            return loss, (loss, ce), res
        (o, r_out, n_out), dis, z = res
        loss, ce = ns["loss_function"](o, td, dis, step, beta=args["beta"])
        la_r, la_n = ns["adversarial_loss"](step, r_out, n_out, rd_arg, nd_arg)
        return loss + la_r + la_n, (loss + la_r + la_n, ce, la_r, la_n), res

    # ---- seeded forward + raw gradients ------------------------------------------------------------------------------
    for step in ((20000, 500) if tag == "fader" else (20000,)):
        for p in model.parameters():
            p.grad = None
        torch.manual_seed(99)
        loss, tup, res = total_loss(step)
        loss.backward()
        S = "%d" % step
        out[P + "loss_terms_" + S] = np.array([float(x) for x in tup])
        sq = 0.0
        for k, p in model.named_parameters():
            if p.grad is not None:
                out[P + "grad_%s/%s" % (S, k)] = p.grad.numpy().copy()
                sq += float((p.grad.double() ** 2).sum())
        out[P + "gradnorm_" + S] = np.array([sq ** 0.5])
        if step == 20000:
            if tag == "fader":
                (o, r_out, n_out), dis, z = res
                out[P + "fw_r_out"], out[P + "fw_n_out"] = r_out.detach().numpy(), n_out.detach().numpy()
            else:
                o, dis, z = res
            out[P + "fw_out"], out[P + "fw_mu"], out[P + "fw_sigma"], out[P + "fw_z"] = (t.detach().numpy() for t in (o, dis.mean, dis.stddev, z))
            out[P + "no_grad_params"] = np.array([k for k, p in model.named_parameters() if p.grad is None])

    # ---- the reference's own train() x3 from step 19999, then evaluate() ---------------------------------------------------
    for p in model.parameters():
        p.grad = None
    ns["optimizer"] = optim.Adam(model.parameters(), lr=args["lr"])
    step, tuples = 19999, []
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = ns["train"](step, d_oh, r_oh, n_oh, td, tr_, tn, tc, rd_arg, nd_arg)
        tuples.append(tup)
    out[P + "train_tuples"] = np.array(tuples, np.float64)
    for k, v in mg.checksums(model.state_dict()).items():
        out[P + "w3sum/" + k] = v
    torch.manual_seed(123)
    if tag == "cvae":
        ev = ns["evaluate"](d_oh, r_oh, n_oh, td, tr_, tn, tc, rd_arg, nd_arg)
    else:
        ev = ns["evaluate"](step - 1, d_oh, r_oh, n_oh, td, tr_, tn, tc, rd_arg, nd_arg)
    out[P + "eval_tuple"] = np.array(ev, np.float64)

    # ---- eval-mode forward on the initial weights (greedy decoder, no rand(1) draws, dropout off) --------------------------------
    model.load_state_dict(sd0)
    model.eval()
    with torch.no_grad():
        torch.manual_seed(7)
        res = call()
    o = res[0][0] if tag == "fader" else res[0]
    top2 = o.topk(2, dim=-1).values
    out[P + "evalfw_tokens"], out[P + "evalfw_gap"] = o.argmax(-1).numpy().astype(np.int16), (top2[..., 0] - top2[..., 1]).numpy()
    out[P + "evalfw_logp0"] = o[:, 0].numpy()
    out[P + "evalfw_z"] = res[2].numpy()
    if tag == "fader":
        out[P + "evalfw_r_out"] = res[0][1].numpy()
    print(tag, "train tuples", np.array(tuples)[:, :2].tolist(), "eval", ev)


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    for tag in CASES:
        make(tag, out)
    path = os.path.join(HERE, "siblings.npz")
    np.savez_compressed(path, **out)
    print("siblings ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
