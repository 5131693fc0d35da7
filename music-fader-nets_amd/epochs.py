"""Epoch driver of the GM-VAE trainer: the schedule, bookkeeping and log lines of the reference's ``training_phase``
(trainer_gmm.py:306-467) on top of GMVAETrainer.

Per epoch: supervised (VGMIDI-style) batches ``(d, r, n, c, arousal, valence, r_density, n_density)`` are trained with
``is_supervised=True, y_label=arousal`` and evaluated (``evaluate`` is called with ``step - 1``, still in train mode, as the reference
does), then the unsupervised (Yamaha-style) batches ``(d, r, n, c, r_density, n_density)``; after each half the per-term means over the
batches are printed in the reference's format, and the state_dict is saved (CPU tensors, reference key set) at the end of every epoch
(trainer_gmm.py:457-459) and once more under a time-stamped name at the end (:461-463).
"""
import os
from datetime import datetime

import torch

HEADER = "D - Data, R - Rhythm, N - Note, RD - Reg. Rhythm, ND- Reg. Note, KLD-L: KLD Latent, KLD-C: KLD Class"
TERMS = "{} loss by term - D: {:.4f} R: {:.4f} N: {:.4f} RD: {:.4f} ND: {:.4f} KLD-L: {:.4f} KLD-C: {:.4f}"


def cpu_state_dict(model):
    """what ``torch.save(model.cpu().state_dict(), path)`` stores, without moving the live model off the GPU"""
    return {k: v.detach().to("cpu", copy=True) for k, v in model.state_dict().items()}


def _run_half(trainer, step, train_dl, val_dl, supervised, log):
    """one half of an epoch (one dataset): train over train_dl, evaluate over val_dl, print the three summary lines"""
    def unpack(x):
        if supervised:
            d, r, n, c, a, _v, r_density, n_density = x
            return (d, r, n, c, r_density, n_density), dict(is_supervised=True, y_label=a)
        d, r, n, c, r_density, n_density = x
        return (d, r, n, c, r_density, n_density), {}

    sums = {"train": [0.0] * 8, "test": [0.0] * 8}
    for x in train_dl:
        (d, r, n, c, rd, nd), kw = unpack(x)
        step, tup = trainer.train(step, None, None, None, d, r, n, c, rd, nd, **kw)
        sums["train"] = [s + float(t) for s, t in zip(sums["train"], tup)]
    for x in val_dl:
        (d, r, n, c, rd, nd), kw = unpack(x)
        tup = trainer.evaluate(step - 1, None, None, None, d, r, n, c, rd, nd, **kw)
        sums["test"] = [s + float(t) for s, t in zip(sums["test"], tup)]
    ntr, nva = len(train_dl), len(val_dl)
    log("batch loss: {:.5f}  {:.5f}".format(sums["train"][0] / ntr, sums["test"][0] / nva))
    log(TERMS.format("train", *[v / ntr for v in sums["train"][1:]]))
    log(TERMS.format("test", *[v / nva for v in sums["test"][1:]]))
    return step


def training_phase(trainer, step, n_epochs, vgm_train_dl, vgm_val_dl, train_dl, val_dl, save_path, name="model", log=print, save=True):
    """Returns the step counter after n_epochs.  The loaders are any sized iterables of batches (lists, DataLoaders).
    save=False: this process does not write checkpoints (data-parallel ranks other than 0)."""
    model = trainer.model
    log(HEADER)
    for i in range(1, n_epochs + 1):
        log("Epoch {} / {}".format(i, n_epochs))
        step = _run_half(trainer, step, vgm_train_dl, vgm_val_dl, True, log)
        step = _run_half(trainer, step, train_dl, val_dl, False, log)
        log("Saving model...")
        if save:
            torch.save(cpu_state_dict(model), save_path)
    if save:
        stamped = os.path.join(os.path.dirname(save_path) or ".", "{}_{}.pt".format(name, datetime.now()))
        torch.save(cpu_state_dict(model), stamped)
    log("Model saved as {}!".format(save_path))
    return step
