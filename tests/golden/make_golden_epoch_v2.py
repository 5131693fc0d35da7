#!/usr/bin/env python3
"""Golden fixture for the entry drivers of the five ``model_config_v2.json`` trainers: ``epoch_v2.npz``.

Runs ONLY in the build container (imports /root/reference).  For every family the reference's OWN ``training_phase`` (and the
``train / evaluate / loss functions`` it calls) is AST-extracted from the run-on-import trainer script and executed unmodified for two
epochs on tiny synthetic Yamaha-style loaders (2 training batches + 1 validation batch of ``(d, r, n, c, r_density, n_density)`` as the
``DataLoader`` collates them: int64 tokens, float32 chroma, float64 densities); recorded per family, prefix ``<tag>/``: the batches, the lines
it printed and the ``state_dict`` it saved to ``params/<name>.pt``.

    tag        script                  model                 start step   what the window crosses
    vae        trainer.py              MusicAttrRegVAE       0            (its beta0 reads a module-level step that never advances)
    singlevae  trainer_singlevae.py    MusicAttrSingleVAE    9998         step 10000
    cvae       trainer_cvae.py         MusicAttrCVAE         9998         step 10000 (evaluate re-derives the densities, reads step 0)
    fader      trainer_fader.py        MusicAttrFaderNets    1998         end of the adversarial ramp (step 2000)
    glsr       trainer_glsr.py         MusicAttrRegVAE       19           the ``step > 20`` gate of the GLSR regulariser (output layer rescaled as
                                                                          in make_golden_glsr.py so that the density walk takes every branch)

Also copies the reference's ``model_config_v2.json`` next to this file (a data file, like ``gmm_model_config.json``).

    python tests/golden/make_golden_epoch_v2.py
"""
import ast
import contextlib
import io
import os
import shutil
import sys
import tempfile
from collections import Counter
from datetime import datetime

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (.cuda() shim, sys.path for the reference)
import make_golden_glsr as gl  # noqa: E402
import model_v2 as ref_v2  # noqa: E402

H, Z = 64, 32
CASES = {  # tag -> (script, class, start step, (B, T, Tr))
    "vae": ("trainer.py", "MusicAttrRegVAE", 0, (6, 20, 8)),
    "singlevae": ("trainer_singlevae.py", "MusicAttrSingleVAE", 9998, (6, 20, 8)),
    "cvae": ("trainer_cvae.py", "MusicAttrCVAE", 9998, (6, 20, 8)),
    "fader": ("trainer_fader.py", "MusicAttrFaderNets", 1998, (6, 20, 8)),
    "glsr": ("trainer_glsr.py", "MusicAttrRegVAE", 19, (4, 104, 8)),
}


def loaders(seed, B, T, TR):
    rng = np.random.RandomState(seed)

    def batch():
        d, r, n, c, rd, nd, _ = mg.synth_batch(rng, B, T, TR)
        return tuple(torch.from_numpy(x) for x in (d, r, n, c, rd, nd))
    return [batch(), batch()], [batch()]


def make(tag, out):
    script, cls, start, (B, T, TR) = CASES[tag]
    torch.manual_seed(1234)
    model = getattr(ref_v2, cls)(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T)
    if tag == "glsr":
        shift = torch.zeros(342)
        shift[180:278] = gl.SEP_SHIFT
        shift[2:90] = gl.NOTE_SHIFT
        with torch.no_grad():
            model.linear_out_g.weight.mul_(gl.OUT_SCALE)
            model.linear_out_g.bias.add_(shift)
        out[tag + "/bias_shift"], out[tag + "/out_scale"] = shift.numpy(), np.array([gl.OUT_SCALE])
    model.train()
    args = {"beta": 0.2, "lr": 1e-3, "n_epochs": 2, "name": "golden_" + tag}
    tr_dl, va_dl = loaders(21, B, T, TR)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "params"))
    save_path = os.path.join(tmp, "params", args["name"] + ".pt")
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, Counter=Counter, model=model, args=args,
              optimizer=mg.optim.Adam(model.parameters(), lr=args["lr"]), tqdm=lambda it, total=None: it, datetime=datetime,
              EVENT_DIMS=342, RHYTHM_DIMS=3, NOTE_DIMS=16, save_path=save_path, train_dl_dist=tr_dl, val_dl_dist=va_dl,
              step=0)                      # the module-level ``step, pre_epoch = 0, 0`` (trainer.py:56) that vae / cvae read
    body = [n for n in ast.parse(open(os.path.join(mg.REF, script)).read()).body
            if isinstance(n, ast.FunctionDef) and n.name != "evaluation_phase"]
    exec(compile(ast.Module(body=body, type_ignores=[]), script + "[extract]", "exec"), ns)
    cwd = os.getcwd()
    os.chdir(tmp)
    buf = io.StringIO()
    try:
        torch.manual_seed(4242)
        with contextlib.redirect_stdout(buf):
            ns["training_phase"](start)
    finally:
        os.chdir(cwd)
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    saved = torch.load(save_path)
    P = tag + "/"
    out[P + "dims"] = np.array([H, Z, B, T, TR])
    out[P + "lines"], out[P + "start_step"] = np.array(lines), np.array(start)
    for name, dl in (("tr", tr_dl), ("va", va_dl)):
        for i, x in enumerate(dl):
            for j, t in enumerate(x):
                out[P + "%s%d_%d" % (name, i, j)] = t.numpy()
    for k, v in saved.items():
        out[P + "wend/" + k] = v.numpy()
    out[P + "stamped"] = np.array(len([f for f in os.listdir(os.path.join(tmp, "params")) if f.startswith(args["name"] + "_")]))
    print("==", tag)
    print("\n".join(lines))


if __name__ == "__main__":
    torch.set_num_threads(8)
    out = {}
    for tag in CASES:
        make(tag, out)
    path = os.path.join(HERE, "epoch_v2.npz")
    np.savez_compressed(path, **out)
    shutil.copyfile(os.path.join(mg.REF, "model_config_v2.json"), os.path.join(HERE, "model_config_v2.json"))
    print("epoch_v2 ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
