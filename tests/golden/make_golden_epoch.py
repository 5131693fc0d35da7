#!/usr/bin/env python3
"""Golden fixture for the epoch driver: runs the reference's OWN ``training_phase``
(trainer_gmm.py:306-467, AST-extracted and executed unmodified, as make_golden.py does for ``train``/``evaluate``) for two epochs on
tiny synthetic "VGMIDI" (supervised) and "Yamaha" (unsupervised) loaders and records what it prints and what it saves.

Runs ONLY in the build container (imports /root/reference).  Output: epoch.npz next to this file.

    python tests/golden/make_golden_epoch.py
"""
import ast
import contextlib
import io
import os
import sys
import tempfile
from datetime import datetime

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the .cuda() shim and imports the reference model)

H, Z, B, T, TR = 64, 32, 6, 20, 8


def loaders(seed):
    """2 supervised train batches + 1 val, 2 unsupervised train batches + 1 val (tuples laid out as the reference unpacks them)."""
    rng = np.random.RandomState(seed)
    def batch(sup):
        d, r, n, c, rd, nd, a = mg.synth_batch(rng, B, T, TR)
        t = [torch.from_numpy(d), torch.from_numpy(r), torch.from_numpy(n), torch.from_numpy(c)]
        return tuple(t + ([torch.from_numpy(a), torch.zeros(B)] if sup else []) + [rd, nd])
    return [batch(True), batch(True)], [batch(True)], [batch(False), batch(False)], [batch(False)]


def main():
    model = mg.build(H, Z)
    model.train()
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    args = {"beta": 0.2, "lr": 1e-3, "n_epochs": 2, "name": "golden"}
    vt, vv, yt, yv = loaders(7)
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, "params"))
    save_path = os.path.join(tmp, "params", "golden.pt")
    ns = dict(torch=torch, np=np, nn=mg.nn, F=mg.F, kl_divergence=mg.kl_divergence, Normal=mg.Normal, model=model, args=args,
              optimizer=mg.optim.Adam(model.parameters(), lr=args["lr"]), tqdm=lambda it, total=None: it, datetime=datetime,
              EVENT_DIMS=342, RHYTHM_DIMS=3, NOTE_DIMS=16, save_path=save_path,
              vgm_train_dl_dist=vt, vgm_val_dl_dist=vv, train_dl_dist=yt, val_dl_dist=yv)
    src = open(os.path.join(mg.REF, "trainer_gmm.py")).read()
    body = [n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in mg.WANTED | {"training_phase"}]
    exec(compile(ast.Module(body=body, type_ignores=[]), "trainer_gmm.py[extract]", "exec"), ns)
    cwd = os.getcwd()
    os.chdir(tmp)
    buf = io.StringIO()
    try:
        torch.manual_seed(4242)
        with contextlib.redirect_stdout(buf):
            ns["training_phase"](9996)          # crosses step 10000: the beta schedule changes sign inside the run
    finally:
        os.chdir(cwd)
    lines = [l for l in buf.getvalue().splitlines() if l.strip()]
    saved = torch.load(save_path)
    out = {"meta_dims": np.array([H, Z, 2, B, T, TR]), "lines": np.array(lines), "start_step": np.array(9996)}
    for name, dl in (("vt", vt), ("vv", vv), ("yt", yt), ("yv", yv)):
        for i, x in enumerate(dl):
            for j, t in enumerate(x):
                out["%s%d_%d" % (name, i, j)] = t.numpy() if torch.is_tensor(t) else np.asarray(t)
    # initial weights: the seeded construction (torch.manual_seed(1234)), identical to small.npz "w0/"
    for k, v in saved.items():
        out["wend/" + k] = v.numpy()
    path = os.path.join(HERE, "epoch.npz")
    np.savez_compressed(path, **out)
    print("epoch ->", path, "%.2f MB" % (os.path.getsize(path) / 1e6))
    print("\n".join(lines))


if __name__ == "__main__":
    torch.set_num_threads(8)
    main()
