"""Shared helpers for the parity tests: golden loading, model construction, comparisons."""
import os

import numpy as np
import torch

from mfn_import import load_package
from oracle import gmvae_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NOISE_PARAMS = ("linear_out_r.bias", "linear_out_n.bias")     # zero-gradient parameters, see test_oracle_golden.py


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))


def sd_from(g, pfx):
    return {k[len(pfx):]: torch.from_numpy(v) for k, v in g.items() if k.startswith(pfx)}


def batch_of(g):
    return {k: g[k] for k in ("d", "r", "n", "c", "r_density", "n_density", "a")}


def make_model(hidden, zdim, sd=None, device="cpu", ops=None, seed=1234):
    pkg = load_package()
    torch.manual_seed(seed)
    m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=hidden, z_dims=zdim,
                              n_step=32, n_component=2)
    if sd is not None:
        m.load_state_dict(sd)
    m = m.to(device)
    if ops is not None:
        m._ops_override = ops
    m.train()
    return m


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))
