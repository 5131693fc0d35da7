#!/usr/bin/env python3
"""Golden fixtures for the EVAL-side callers of the path: ``eval.npz``.

Runs ONLY in the build container.  The reference's own code is executed, not restated:
  * ``gmm_model.MusicAttrRegGMVAE`` imported with the ``.cuda()`` shim (make_golden.py);
  * ``convert_to_one_hot, clean_output, repar, BaseEvaluator, RhythmEvaluator, NoteEvaluator`` AST-extracted from
    ``test_class.py`` and ``GMMRhythmEvaluator, GMMNoteEvaluator, run_through_gmm`` from ``test_gmm_v2.py`` (both files are
    scripts that import the MIDI stack, so they cannot be imported);
  * the code of ``arousal_transfer.ipynb`` cells 11 and 15 / 17 (up to the ``global_decoder`` call) executed as written.

Sections (for each model size: "s" = hidden 64 / z 32, "c" = hidden 512 / z 128 = BASELINE config 0 dims)
  A  eval-mode ``model(d_oh, r_oh, n_oh, c)``: the decoder feeds back its own argmax (gmm_model.py:146-148)
  B  ``GMMRhythmEvaluator.shift`` / ``GMMNoteEvaluator.shift`` (test_class.py:233-254, :282-303): first call in train mode
     (as the reference's evaluator finds the model), later calls in eval mode (nobody resets it, :250)
  C  notebook transfer: z + lambda * (mu_lookup[1] - mu_lookup[0]) on BOTH latents, 300 greedy steps
  D  ``run_through_gmm`` over a small loader (forward only; min / max of z[:, 0])
Every record stores the torch seed set right before the call; tokens, top-2 gaps and the first log-prob row are stored.
"""
import ast
import json
import os
import sys
from collections import Counter

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
from torch.distributions import Normal  # noqa: E402

REF = mg.REF


def extract(path, names, ns):
    tree = ast.parse(open(os.path.join(REF, path)).read())
    body = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert {n.name for n in body} == set(names), (path, names)
    exec(compile(ast.Module(body=body, type_ignores=[]), path + "[extract]", "exec"), ns)


def notebook_cells():
    nb = json.load(open(os.path.join(REF, "arousal_transfer.ipynb")))
    cells = {i: "".join(c["source"]) for i, c in enumerate(nb["cells"]) if c["cell_type"] == "code"}

    def head(src):              # up to and including the global_decoder call (the rest needs the MIDI stack / IPython)
        lines = src.split("\n")
        k = max(i for i, l in enumerate(lines) if "model.global_decoder" in l)
        return "\n".join(lines[:k + 1])
    return cells[11], head(cells[15]), head(cells[17])


def dec_record(out):
    tok = out.argmax(-1)
    top2 = out.topk(2, dim=-1).values
    return tok.numpy().astype(np.int16), (top2[..., 0] - top2[..., 1]).numpy()


def make(tag, hidden, zdim, B, T, Tr, out):
    model = mg.build(hidden, zdim)
    ns = dict(torch=torch, np=np, Normal=Normal, Counter=Counter, model=model, tqdm=lambda x, **k: x,
              EVENT_DIMS=342, RHYTHM_DIMS=3, NOTE_DIMS=16, CHROMA_DIMS=24)
    extract("test_class.py", {"convert_to_one_hot", "clean_output", "repar", "BaseEvaluator", "RhythmEvaluator", "NoteEvaluator"}, ns)
    extract("test_gmm_v2.py", {"GMMRhythmEvaluator", "GMMNoteEvaluator", "run_through_gmm"}, ns)
    d, r, n, c, r_den, n_den, _ = mg.synth_batch(np.random.RandomState(3), B, T, Tr)
    td, tr_, tn, tc = torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)
    P = tag + "/"
    out[P + "dims"] = np.array([hidden, zdim, 2, B, T, Tr])
    for k, v in (("d", d), ("r", r), ("n", n), ("c", c), ("r_density", r_den), ("n_density", n_den)):
        out[P + k] = v
    for k, v in mg.checksums(model.state_dict()).items():
        out[P + "w0sum/" + k] = v

    with torch.no_grad():
        # ---- D: run_through_gmm (train mode as constructed, forward only) -------------------------------------------
        dl = [(td[i:i + 3], tr_[i:i + 3], tn[i:i + 3], tc[i:i + 3], torch.from_numpy(r_den[i:i + 3]), torch.from_numpy(n_den[i:i + 3]))
              for i in range(0, B, 3)]
        torch.manual_seed(5)
        res = ns["run_through_gmm"](dl)
        names = ["r_density_lst", "n_density_lst", "r_lst", "n_lst", "a_lst", "r_mean", "n_mean", "z_r_0_lst", "z_r_rest_lst",
                 "z_n_0_lst", "z_n_rest_lst", "r_min", "r_max", "n_min", "n_max"]
        for k, v in zip(names, res):
            if k != "a_lst":
                out[P + "rt_" + k] = np.asarray(v)

        # ---- B: evaluator shifts: call 0 finds the model in train mode, it stays in eval mode afterwards ------------
        model.train()
        rv, nv = ns["GMMRhythmEvaluator"](None), ns["GMMNoteEvaluator"](None)
        calls = [("r", 0, 0.75), ("r", 1, -1.5), ("n", 1, 0.4), ("n", 2, 2.0), ("r", 2, 0.0)]
        for k, (which, i, val) in enumerate(calls):
            out[P + "shift%d_training_before" % k] = np.array([int(model.training)])
            torch.manual_seed(100 + k)
            o, z0 = (rv if which == "r" else nv).shift(model, td[i], tr_[i], tn[i], tc[i], val)
            tok, gap = dec_record(o)
            out[P + "shift%d_tokens" % k], out[P + "shift%d_gap" % k] = tok, gap
            out[P + "shift%d_z0" % k] = np.array([z0])
            out[P + "shift%d_logp0" % k] = o[0, 0].numpy()
            out[P + "shift%d_clean" % k] = np.asarray(ns["clean_output"](o)).astype(np.int16)
        out[P + "shift_calls"] = np.array([[{"r": 0, "n": 1}[w], i, v] for w, i, v in calls], np.float64)

        # ---- A: eval-mode forward ------------------------------------------------------------------------------------
        model.eval()
        d_oh, r_oh, n_oh = (ns["convert_to_one_hot"](x, v) for x, v in ((td, 342), (tr_, 3), (tn, 16)))
        torch.manual_seed(7)
        (o, r_out, n_out, _, _), dis, z_out, ll_out, qy_out, y_out = model(d_oh, r_oh, n_oh, tc)
        tok, gap = dec_record(o)
        out[P + "evalfw_tokens"], out[P + "evalfw_gap"] = tok, gap
        out[P + "evalfw_logp0"] = o[:, 0].numpy()
        for k, v in dict(r_out=r_out, n_out=n_out, mu_r=dis[0].mean, sigma_r=dis[0].stddev, z_r=z_out[0], z_n=z_out[1],
                         ll_r=ll_out[0], qy_n=qy_out[1]).items():
            out[P + "evalfw_" + k] = v.numpy()
        out[P + "evalfw_y_r"], out[P + "evalfw_y_n"] = y_out[0].numpy(), y_out[1].numpy()
        after = torch.rand(1).item()          # position of the generator after the call (eval mode draws no rand(1) per step)
        torch.manual_seed(7)
        torch.randn(B, zdim), torch.randn(B, zdim)
        assert after == torch.rand(1).item()

        # ---- C: notebook cells 11 + 15 (low -> high) / 17 (high -> low), executed as written ---------------------------
        cell11, cell15, cell17 = notebook_cells()
        for j, (i, cell, seed) in enumerate(((0, cell15, 31), (1, cell17, 32))):
            nb = dict(ns)
            nb.update(d_oh=ns["convert_to_one_hot"](td[i], 342), c=tc[i].unsqueeze(0))
            exec(cell11, nb)
            torch.manual_seed(seed)
            exec(cell, nb)
            tok, gap = dec_record(nb["out"])
            out[P + "nb%d_tokens" % j], out[P + "nb%d_gap" % j] = tok, gap
            out[P + "nb%d_z" % j] = nb["z"].numpy()
            out[P + "nb%d_clean" % j] = np.asarray(ns["clean_output"](nb["out"])).astype(np.int16)
            out[P + "nb%d_meta" % j] = np.array([i, seed, nb["lmbda"], nb["out"].shape[1]])
    print(tag, "done")


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    make("s", 64, 32, 6, 20, 8, out)
    make("c", 512, 128, 6, 64, 16, out)
    path = os.path.join(HERE, "eval.npz")
    np.savez_compressed(path, **out)
    print("eval ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
