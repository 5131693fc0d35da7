"""Entry driver of the GM-VAE trainer: the module body of the reference's ``trainer_gmm.py`` (:21-96 config / model / optimiser /
loaders, :306-467 ``training_phase``, :613 the call) on the HIP path.

    python train_gmm.py --config gmm_model_config.json [--synthetic | --data-root DIR] [--epochs N] [--out DIR]

* the config is the reference's JSON schema, read verbatim: ``batch_size, n_epochs, lr, decay, name, hidden_dim, z_dim, beta, time_step,
  num_clusters`` (``gmm_model_config.json`` / ``model_config_v2.json`` load unchanged; ``decay`` is read and - as in the reference, which
  never builds a scheduler from it - unused);
* ``params/<name>.pt`` is loaded when it exists ("Loading ..."), written after every epoch and once more under a time-stamped name at
  the end; ``log/`` and ``params/`` are created like the reference does;
* data: ``--data-root DIR`` reads the reference's pre-processed arrays (``DIR/values_v3/*.npy``, ``DIR/filtered_songs_disambiguate/*.npy``,
  ptb_v2.py:344-397); ``--synthetic`` builds arrays of the same structure from a seeded generator (no dataset ships with this repository,
  and the MIDI tokeniser is out of scope).  Yamaha loaders use ``batch_size``, VGMIDI loaders 32 (trainer_gmm.py:69,84), ``shuffle=True``;
* under ``torch.distributed.run`` every rank trains on its rows of each batch (data parallel, parallel.py); rank 0 prints and saves.
"""
import argparse
import json
import os
import sys

import numpy as np
import torch
from torch.utils.data import DataLoader

CONFIG_KEYS = ("batch_size", "n_epochs", "lr", "decay", "name", "hidden_dim", "z_dim", "beta", "time_step", "num_clusters")
EVENT_DIMS, RHYTHM_DIMS, NOTE_DIMS, CHROMA_DIMS = 342, 3, 16, 24           # trainer_gmm.py:35-38


def read_config(path):
    with open(path) as f:
        args = json.load(f)
    missing = [k for k in CONFIG_KEYS if k not in args and k != "decay"]
    if missing:
        raise KeyError("config %s lacks %s (the reference's schema: %s)" % (path, missing, ", ".join(CONFIG_KEYS)))
    return args


def synthetic_arrays(n_yamaha, n_vgmidi, T, Tr, seed=0):
    """arrays with the structure of the reference's caches: dense Yamaha arrays, ragged VGMIDI object arrays"""
    from .synth import synth_batch
    rng = np.random.RandomState(seed)
    y = synth_batch(rng, n_yamaha, T, Tr)
    yam = (y["d"].astype(np.float32), y["r"].astype(np.float32), y["n"].astype(np.float32), y["c"])
    v = synth_batch(rng, n_vgmidi, T, Tr)
    obj = lambda rows: np.array([np.asarray(r) for r in rows] + [None], dtype=object)[:-1]
    lens = rng.randint(T // 2, T - 1, size=n_vgmidi)
    data = obj([v["d"][i, :lens[i]] for i in range(n_vgmidi)])
    rl = rng.randint(Tr // 2, Tr + 1, size=n_vgmidi)
    rhythm = obj([v["r"][i, :rl[i]] for i in range(n_vgmidi)])
    note = obj([v["n"][i, :rl[i]] for i in range(n_vgmidi)])
    arousal = rng.uniform(-1, 1, size=n_vgmidi)
    valence = rng.uniform(-1, 1, size=n_vgmidi)
    return yam, (data, rhythm, note, arousal, valence, v["c"])


class RankShard:
    """this rank's rows of every batch of a loader (equal shards, the ragged remainder of a batch is dropped)"""

    def __init__(self, dl, rank, world):
        self.dl, self.rank, self.world = dl, rank, world

    def __len__(self):
        return len(self.dl)

    def __iter__(self):
        for x in self.dl:
            per = len(x[0]) // self.world
            if per == 0:
                continue
            lo = self.rank * per
            yield [t[lo:lo + per] for t in x]


def build_loaders(args, opts, rank, world):
    from . import datasets as D
    if opts.data_root:
        yam = D.load_yamaha_arrays(os.path.join(opts.data_root, "values_v3"))
        vgm = D.load_vgmidi_arrays(os.path.join(opts.data_root, "filtered_songs_disambiguate"))
    else:
        yam, vgm = synthetic_arrays(opts.synthetic_songs, max(64, opts.synthetic_songs // 4), opts.seq_len, max(8, opts.seq_len // 4))
    data, rhythm, note, chroma = yam
    vd, vr, vn, va, vv, vc = vgm
    mk = lambda ds, bs: DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0)
    dls = dict(train=mk(D.YamahaDataset(data, rhythm, note, chroma, mode="train"), args["batch_size"]),
               val=mk(D.YamahaDataset(data, rhythm, note, chroma, mode="val"), args["batch_size"]),
               vgm_train=mk(D.VGMIDIDataset(vd, vr, vn, vc, np.array(va, copy=True), vv, mode="train"), 32),
               vgm_val=mk(D.VGMIDIDataset(vd, vr, vn, vc, np.array(va, copy=True), vv, mode="val"), 32))
    sizes = {k: len(v.dataset) for k, v in dls.items()}
    if world > 1:
        dls = {k: RankShard(v, rank, world) for k, v in dls.items()}
    return dls, sizes


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True, help="the reference's JSON config (gmm_model_config.json schema)")
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--synthetic", action="store_true", help="seeded synthetic datasets (default when no --data-root is given)")
    g.add_argument("--data-root", help="directory holding values_v3/ and filtered_songs_disambiguate/ (the reference's data/)")
    ap.add_argument("--synthetic-songs", type=int, default=512)
    ap.add_argument("--seq-len", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=None, help="override n_epochs of the config")
    ap.add_argument("--out", default=".", help="where log/ and params/ live (default: the working directory, as the reference)")
    ap.add_argument("--seed", type=int, default=None)
    opts = ap.parse_args(argv)

    from . import GMVAETrainer, MusicAttrRegGMVAE, parallel
    from .epochs import training_phase
    args = read_config(opts.config)
    if not torch.cuda.is_available():
        raise SystemExit("train: needs an MI355X - the HIP path has no CPU fallback")
    ctx, local = parallel.init_from_env()
    rank, world = (0, 1) if ctx is None else (ctx.rank, ctx.world)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    say = print if rank == 0 else (lambda *a, **k: None)
    for sub in ("log", "params"):                                   # trainer_gmm.py:23-26
        os.makedirs(os.path.join(opts.out, sub), exist_ok=True)
    save_path = os.path.join(opts.out, "params", "{}.pt".format(args["name"]))

    if opts.seed is not None:
        torch.manual_seed(opts.seed)
    model = MusicAttrRegGMVAE(roll_dims=EVENT_DIMS, rhythm_dims=RHYTHM_DIMS, note_dims=NOTE_DIMS, chroma_dims=CHROMA_DIMS,
                              hidden_dims=args["hidden_dim"], z_dims=args["z_dim"], n_step=args["time_step"], n_component=args["num_clusters"])
    if os.path.exists(save_path):
        say("Loading {}".format(save_path))
        model.load_state_dict(torch.load(save_path, map_location="cpu"))
    else:
        say("Save path: {}".format(save_path))
    say("Using: ", torch.cuda.get_device_name(dev))
    model.to(dev)
    trainer = GMVAETrainer(model, lr=args["lr"], beta=args["beta"], dist_ctx=ctx)

    say("Loading Yamaha..." if opts.data_root else "Building synthetic Yamaha / VGMIDI arrays...")
    dls, sizes = build_loaders(args, opts, rank, world)
    say("Yamaha: Train / Validation")
    say(sizes["train"], sizes["val"])
    say("VGMIDI: Train / Validation")
    say(sizes["vgm_train"], sizes["vgm_val"])
    say()

    n_epochs = args["n_epochs"] if opts.epochs is None else opts.epochs
    step = training_phase(trainer, 0, n_epochs, dls["vgm_train"], dls["vgm_val"], dls["train"], dls["val"], save_path,
                          name=args["name"], log=say, save=rank == 0)        # every rank runs the schedule, rank 0 writes the checkpoints
    if ctx is not None:
        torch.distributed.barrier()
        if ctx.rccl is not None:
            ctx.rccl.close()
        torch.distributed.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
