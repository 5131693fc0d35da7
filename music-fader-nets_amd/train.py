"""Entry driver of the six reference trainers: their module bodies (config / model / optimiser / loaders, then ``training_phase``) on the
HIP path.

    python train_gmm.py --config gmm_model_config.json [--model gmm] [--synthetic | --data-root DIR] [--epochs N] [--out DIR]
    python train_gmm.py --config model_config_v2.json --model {vae,singlevae,cvae,fader,glsr} ...

    --model      reference script             model class            trainer             config the script opens
    gmm          trainer_gmm.py:21-96,613     MusicAttrRegGMVAE      GMVAETrainer        gmm_model_config.json   (VGMIDI + Yamaha loaders)
    vae          trainer.py:20-76,380         MusicAttrRegVAE        VAETrainer          model_config_v2.json    (Yamaha loaders only)
    singlevae    trainer_singlevae.py:19-75   MusicAttrSingleVAE     SingleVAETrainer    model_config_v2.json
    cvae         trainer_cvae.py:17-73        MusicAttrCVAE          CVAETrainer         model_config_v2.json
    fader        trainer_fader.py:17-73       MusicAttrFaderNets     FaderTrainer        model_config_v2.json
    glsr         trainer_glsr.py:18-75        MusicAttrRegVAE        GLSRTrainer         model_config_v2.json    (single process)

* the config is the reference's JSON schema, read verbatim: ``batch_size, n_epochs, lr, decay, name, hidden_dim, z_dim, beta, time_step``
  and, for the GM-VAE only, ``num_clusters`` - ``gmm_model_config.json`` and ``model_config_v2.json`` load unchanged (``decay`` is read
  and - as in the reference, which never builds a scheduler from it - unused);
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

CONFIG_KEYS = ("batch_size", "n_epochs", "lr", "decay", "name", "hidden_dim", "z_dim", "beta", "time_step")      # model_config_v2.json
GMM_KEYS = ("num_clusters",)                                               # + gmm_model_config.json (trainer_gmm.py:43)
EVENT_DIMS, RHYTHM_DIMS, NOTE_DIMS, CHROMA_DIMS = 342, 3, 16, 24           # trainer_gmm.py:35-38
MODELS = {   # --model -> (model class, trainer class) of this package
    "gmm": ("MusicAttrRegGMVAE", "GMVAETrainer"), "vae": ("MusicAttrRegVAE", "VAETrainer"), "glsr": ("MusicAttrRegVAE", "GLSRTrainer"),
    "singlevae": ("MusicAttrSingleVAE", "SingleVAETrainer"), "cvae": ("MusicAttrCVAE", "CVAETrainer"), "fader": ("MusicAttrFaderNets", "FaderTrainer"),
}


def read_config(path, model="gmm"):
    """the reference's JSON config as the script of `model` reads it: the keys that script indexes must be there (``num_clusters`` only
    for the GM-VAE), anything else in the file is carried along untouched"""
    with open(path) as f:
        args = json.load(f)
    need = CONFIG_KEYS + (GMM_KEYS if model == "gmm" else ())
    missing = [k for k in need if k not in args and k != "decay"]
    if missing:
        raise KeyError("config %s lacks %s (the schema --model %s reads: %s)" % (path, missing, model, ", ".join(need)))
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

    may_skip = True                  # epochs._count: average over the batches actually yielded

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


def build_loaders(args, opts, rank, world, model="gmm", seed=0):
    """the loaders of the script of `model`: Yamaha train / val with ``batch_size`` (+ the test split's size, which the v2 scripts print),
    and for the GM-VAE the VGMIDI pair with batch size 32 (trainer_gmm.py:69,84).  ``shuffle=True`` draws its permutations from an
    explicit generator seeded with `seed` - under data parallelism every rank must see the SAME batches to take its rows of them, whatever
    else has consumed the global generator on a rank."""
    from . import datasets as D
    if opts.data_root:
        yam = D.load_yamaha_arrays(os.path.join(opts.data_root, "values_v3"))
        vgm = D.load_vgmidi_arrays(os.path.join(opts.data_root, "filtered_songs_disambiguate")) if model == "gmm" else None
    else:
        yam, vgm = synthetic_arrays(opts.synthetic_songs, max(64, opts.synthetic_songs // 4), opts.seq_len, max(8, opts.seq_len // 4))
    data, rhythm, note, chroma = yam

    def mk(ds, bs, k):
        g = torch.Generator()
        g.manual_seed(seed * 8 + k)
        return DataLoader(ds, batch_size=bs, shuffle=True, num_workers=0, generator=g)

    dls = dict(train=mk(D.YamahaDataset(data, rhythm, note, chroma, mode="train"), args["batch_size"], 0),
               val=mk(D.YamahaDataset(data, rhythm, note, chroma, mode="val"), args["batch_size"], 1))
    sizes = {k: len(v.dataset) for k, v in dls.items()}
    sizes["test"] = len(D.YamahaDataset(data, rhythm, note, chroma, mode="test"))
    if model == "gmm":
        vd, vr, vn, va, vv, vc = vgm
        dls["vgm_train"] = mk(D.VGMIDIDataset(vd, vr, vn, vc, np.array(va, copy=True), vv, mode="train"), 32, 2)
        dls["vgm_val"] = mk(D.VGMIDIDataset(vd, vr, vn, vc, np.array(va, copy=True), vv, mode="val"), 32, 3)
        sizes.update(vgm_train=len(dls["vgm_train"].dataset), vgm_val=len(dls["vgm_val"].dataset))
    if world > 1:
        dls = {k: RankShard(v, rank, world) for k, v in dls.items()}
    return dls, sizes


def shared_seed(opts, ctx):
    """one seed for every rank: --seed, else rank 0's draw.  Data parallelism relies on lockstep CPU generators - the reparameterisation
    noise is drawn for the GLOBAL batch and sliced per rank (GMVAETrainer.draw_eps), the loaders shuffle identically."""
    seed = opts.seed if opts.seed is not None else int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
    if ctx is not None and ctx.world > 1:
        box = [seed]
        torch.distributed.broadcast_object_list(box, src=0, group=ctx.group)
        seed = int(box[0])
    return seed


def sync_replicas(trainer, ctx):
    """every replica starts from rank 0's weights (a rank that missed ``params/<name>.pt`` on node-local storage would otherwise train a
    different model for ever: only gradients are exchanged).  SUM all-reduce with the other ranks' copies zeroed = a broadcast through the
    one collective the data plane has."""
    if ctx is None or ctx.world == 1:
        return
    flat = trainer.flat.param
    with torch.no_grad():
        if ctx.rank != 0:
            flat.zero_()
        ctx.all_reduce_sum(flat)
    if flat.is_cuda:
        torch.cuda.synchronize()
    trainer.model.weights_changed()


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", required=True, help="the reference's JSON config (gmm_model_config.json / model_config_v2.json)")
    ap.add_argument("--model", choices=sorted(MODELS), default="gmm", help="which reference trainer script to run (default: trainer_gmm.py)")
    g = ap.add_mutually_exclusive_group()
    g.add_argument("--synthetic", action="store_true", help="seeded synthetic datasets (default when no --data-root is given)")
    g.add_argument("--data-root", help="directory holding values_v3/ and filtered_songs_disambiguate/ (the reference's data/)")
    ap.add_argument("--synthetic-songs", type=int, default=512)
    ap.add_argument("--seq-len", type=int, default=100)
    ap.add_argument("--epochs", type=int, default=None, help="override n_epochs of the config")
    ap.add_argument("--out", default=".", help="where log/ and params/ live (default: the working directory, as the reference)")
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--arith", choices=["f32", "bf16x6"], default=None,
                    help="arithmetic of the deep MFMA products (arith.py): f32 = fp32 MFMA chains; bf16x6 = every fp32 operand cut exactly into three "
                         "bf16 pieces, six partial products on the bf16 MFMA, fp32 accumulation (as accurate as the fp32 chains against float64). "
                         "Default: the package default (arith.default())")
    ap.add_argument("--bf16x6", action="store_true", help="same as --arith bf16x6")
    opts = ap.parse_args(argv)

    import importlib
    pkg = importlib.import_module(__package__)
    from . import parallel
    from .epochs import training_phase, training_phase_v2
    args = read_config(opts.config, opts.model)
    if not torch.cuda.is_available():
        raise SystemExit("train: needs an MI355X - the HIP path has no CPU fallback")
    ctx, local = parallel.init_from_env()
    rank, world = (0, 1) if ctx is None else (ctx.rank, ctx.world)
    if opts.model == "glsr" and world > 1:
        raise SystemExit("train: --model glsr is single-process (the reference's rhythm-density walk reads sample 0 of the batch for every row)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    say = print if rank == 0 else (lambda *a, **k: None)
    for sub in ("log", "params"):                                   # trainer_gmm.py:23-26
        os.makedirs(os.path.join(opts.out, sub), exist_ok=True)
    save_path = os.path.join(opts.out, "params", "{}.pt".format(args["name"]))

    seed = shared_seed(opts, ctx)
    if opts.seed is not None or world > 1:
        torch.manual_seed(seed)
    model_cls, trainer_cls = (getattr(pkg, n) for n in MODELS[opts.model])
    kw = dict(roll_dims=EVENT_DIMS, rhythm_dims=RHYTHM_DIMS, note_dims=NOTE_DIMS, chroma_dims=CHROMA_DIMS,
              hidden_dims=args["hidden_dim"], z_dims=args["z_dim"], n_step=args["time_step"])
    if opts.model == "gmm":
        kw["n_component"] = args["num_clusters"]
    model = model_cls(**kw)
    if os.path.exists(save_path):
        say("Loading {}".format(save_path))
        model.load_state_dict(torch.load(save_path, map_location="cpu"))
    else:
        say("Save path: {}".format(save_path))
    say("Using: ", torch.cuda.get_device_name(dev))
    model.to(dev)
    trainer = trainer_cls(model, lr=args["lr"], beta=args["beta"], **({} if opts.model == "glsr" else {"dist_ctx": ctx}))
    sync_replicas(trainer, ctx)
    if opts.bf16x6 or opts.arith is not None:
        model.set_arith("bf16x6" if opts.bf16x6 else opts.arith)        # lives on the model: survives engine rebuilds (.to(), re-homed parameters)

    say("Loading Yamaha..." if opts.data_root else "Building synthetic Yamaha / VGMIDI arrays...")
    dls, sizes = build_loaders(args, opts, rank, world, opts.model, seed)
    n_epochs = args["n_epochs"] if opts.epochs is None else opts.epochs
    if opts.model == "gmm":
        say("Yamaha: Train / Validation")
        say(sizes["train"], sizes["val"])
        say("VGMIDI: Train / Validation")
        say(sizes["vgm_train"], sizes["vgm_val"])
        say()
        step = training_phase(trainer, 0, n_epochs, dls["vgm_train"], dls["vgm_val"], dls["train"], dls["val"], save_path,
                              name=args["name"], log=say, save=rank == 0)    # every rank runs the schedule, rank 0 writes the checkpoints
    else:
        say("Train / Validation / Test")                                # trainer.py:74-75
        say(sizes["train"], sizes["val"], sizes["test"])
        step = training_phase_v2(opts.model, trainer, 0, n_epochs, dls["train"], dls["val"], save_path, name=args["name"], log=say, save=rank == 0)
    if ctx is not None:
        torch.distributed.barrier()
        if ctx.rccl is not None:
            ctx.rccl.close()
        torch.distributed.destroy_process_group()
    return step


if __name__ == "__main__":
    main()
