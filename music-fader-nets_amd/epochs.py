"""Epoch drivers: the schedule, bookkeeping and log lines of the reference's ``training_phase`` functions - the GM-VAE trainer's
(trainer_gmm.py:306-467, ``training_phase`` below, on top of GMVAETrainer) and those of the five ``model_config_v2.json`` trainers
(``training_phase_v2``: trainer.py:191-271, trainer_singlevae.py:184-255, trainer_cvae.py:156-223, trainer_fader.py:164-236,
trainer_glsr.py:304-379).

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


def _count(dl, seen):
    """divisor of the per-epoch means: the reference divides by ``len(dl)``; a rank shard that had to skip batches smaller than the world
    (train.py: RankShard) yields fewer than it announces - then the batches actually seen count"""
    if getattr(dl, "may_skip", False):
        if seen == 0:
            raise RuntimeError("a data-parallel shard yielded no batch at all: every batch of this loader is smaller than the world size "
                               "(%d batches announced) - use a batch size of at least one row per rank" % len(dl))
        return seen
    return len(dl)


def _run_half(trainer, step, train_dl, val_dl, supervised, log):
    """one half of an epoch (one dataset): train over train_dl, evaluate over val_dl, print the three summary lines"""
    def unpack(x):
        if supervised:
            d, r, n, c, a, _v, r_density, n_density = x
            return (d, r, n, c, r_density, n_density), dict(is_supervised=True, y_label=a)
        d, r, n, c, r_density, n_density = x
        return (d, r, n, c, r_density, n_density), {}

    sums = {"train": [0.0] * 8, "test": [0.0] * 8}
    ntr = nva = 0
    for x in train_dl:
        (d, r, n, c, rd, nd), kw = unpack(x)
        step, tup = trainer.train(step, None, None, None, d, r, n, c, rd, nd, **kw)
        sums["train"] = [s + float(t) for s, t in zip(sums["train"], tup)]
        ntr += 1
    for x in val_dl:
        (d, r, n, c, rd, nd), kw = unpack(x)
        tup = trainer.evaluate(step - 1, None, None, None, d, r, n, c, rd, nd, **kw)
        sums["test"] = [s + float(t) for s, t in zip(sums["test"], tup)]
        nva += 1
    ntr, nva = _count(train_dl, ntr), _count(val_dl, nva)
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


# ---------------------------------------------------------------------------------------------------------------------------------
# the five model_config_v2.json trainers: one Yamaha-style loader pair, six batch fields (d, r, n, c, r_density, n_density)
# ---------------------------------------------------------------------------------------------------------------------------------
HEADER_V2 = "D - Data, R - Rhythm, N - Note, RD - Reg. Rhythm Density, ND- Reg. Note Density"
HEADER_FADER = "D - Data, R - Rhythm, N - Note, RA - Rhythm Adversarial, NA - Note Adversarial"
TERMS_V2 = "{} loss by term - D: {:.4f} R: {:.4f} N: {:.4f} {}: {:.4f} {}: {:.4f}"

# family -> what its reference loop does.  slots: where the entries of the returned tuple go in the printed (D, R, N, x, y) line -
# the reference's loops keep b_CE_R / b_CE_N (and, CVAE, b_l_r / b_l_n) at their initial 0 when the step does not return them;
# eval: how ``evaluate`` is called after the training batches ("step-1", "step" or no step argument);
# col: the densities are handed over as float32 (B, 1) columns (trainer_cvae.py:171-172, trainer_fader.py:180-181);
# each: "Saving model..." + checkpoint after EVERY epoch (otherwise once, behind the last epoch)
FAMILIES = {
    "vae": dict(fmt="{:.4f}", slots=(0, 1, 2, 3, 4), eval="none", col=False, each=True, header=HEADER_V2, tags=("RD", "ND")),
    "glsr": dict(fmt="{:.5f}", slots=(0, 1, 2, 3, 4), eval="step-1", col=False, each=True, header=HEADER_V2, tags=("RD", "ND")),
    "singlevae": dict(fmt="{:.5f}", slots=(0, 3, 4), eval="step-1", col=False, each=True, header=HEADER_V2, tags=("RD", "ND")),
    "cvae": dict(fmt="{:.5f}", slots=(0,), eval="none", col=True, each=False, header=HEADER_V2, tags=("RD", "ND")),
    "fader": dict(fmt="{:.5f}", slots=(0, 3, 4), eval="step", col=True, each=False, header=HEADER_FADER, tags=("RA", "NA")),
}


def training_phase_v2(family, trainer, step, n_epochs, train_dl, val_dl, save_path, name="model", log=print, save=True):
    """``training_phase(step)`` of the reference trainer of `family` ("vae" = trainer.py, "singlevae", "cvae", "fader", "glsr") on the
    matching trainer object of this package.  Returns the step counter after n_epochs."""
    F = FAMILIES[family]
    model = trainer.model

    def fields(x):
        d, r, n, c, rd, nd = x
        if F["col"]:
            rd, nd = torch.as_tensor(rd).float().unsqueeze(-1), torch.as_tensor(nd).float().unsqueeze(-1)
        return d, r, n, c, rd, nd

    def add(acc, tup):
        acc[0] += float(tup[0])
        for slot, v in zip(F["slots"], tup[1:]):
            acc[1 + slot] += float(v)

    log(F["header"])
    for i in range(1, n_epochs + 1):
        log("Epoch {} / {}".format(i, n_epochs))
        tr, te = [0.0] * 6, [0.0] * 6
        n_tr = n_te = 0
        for x in train_dl:
            d, r, n, c, rd, nd = fields(x)
            step, tup = trainer.train(step, None, None, None, d, r, n, c, rd, nd)
            add(tr, tup)
            n_tr += 1
        for x in val_dl:
            d, r, n, c, rd, nd = fields(x)
            if F["eval"] == "none":
                tup = trainer.evaluate(None, None, None, d, r, n, c, rd, nd)
            else:
                tup = trainer.evaluate(step - 1 if F["eval"] == "step-1" else step, None, None, None, d, r, n, c, rd, nd)
            add(te, tup)
            n_te += 1
        n_tr, n_te = _count(train_dl, n_tr), _count(val_dl, n_te)
        log(("batch loss: " + F["fmt"] + "  " + F["fmt"]).format(tr[0] / n_tr, te[0] / n_te))
        log(TERMS_V2.format("train", tr[1] / n_tr, tr[2] / n_tr, tr[3] / n_tr, F["tags"][0], tr[4] / n_tr, F["tags"][1], tr[5] / n_tr))
        log(TERMS_V2.format("test", te[1] / n_te, te[2] / n_te, te[3] / n_te, F["tags"][0], te[4] / n_te, F["tags"][1], te[5] / n_te))
        if F["each"]:
            log("Saving model...")
            if save:
                torch.save(cpu_state_dict(model), save_path)
    if save:
        if not F["each"]:
            torch.save(cpu_state_dict(model), save_path)
        stamped = os.path.join(os.path.dirname(save_path) or ".", "{}_{}.pt".format(name, datetime.now()))
        torch.save(cpu_state_dict(model), stamped)
    log("Model saved as {}!".format(save_path))
    return step
