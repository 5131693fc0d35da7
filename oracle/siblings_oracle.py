"""CPU oracle for the single-encoder siblings  --  TEST INFRASTRUCTURE, NOT PRODUCT (only ``tests/`` may import this file).

From-scratch fp32 restatement of
    MusicAttrSingleVAE  model_v2.py:174-285  + loss / step of trainer_singlevae.py:84-140
    MusicAttrCVAE       model_v2.py:288-423  + trainer_cvae.py:84-135
    MusicAttrFaderNets  model_v2.py:426-586  + trainer_fader.py:84-146
on the GRU / decoder blocks of ``gmvae_oracle``.  Parity pin: ``tests/golden/siblings.npz`` produced by importing the reference itself
(``tests/golden/make_golden_siblings.py``), checked in ``tests/test_oracle_golden.py``.

Quirks restated on purpose: the single-encoder VAE adds ``beta * KLD`` with its CONSTANT beta although it computes an annealed beta0
(trainer_singlevae.py:90-104); its regulariser reads columns 0 and 1 of ONE latent (:107-120); the CVAE's ``evaluate`` re-derives the
densities from the tokens and reads the module-level ``step`` = 0 (trainer_cvae.py:120-133); the Fader heads see ``reverse(z)``, so
the encoder receives MINUS their gradient (model_v2.py:426-435, :572-575).
"""
import math

import numpy as np
import torch
from torch import nn

from . import gmvae_oracle as g

E = g.E
KINDS = ("single", "cvae", "fader")


def init_state_dict(kind, hidden, zdim, seed=1234):
    """``torch.manual_seed(seed); <Model>(...)``: the torch constructors in the reference's order."""
    torch.manual_seed(seed)
    H, Z = hidden, zdim
    if kind == "single":
        mods = [("gru", nn.GRU(E, H, batch_first=True, bidirectional=True)), ("mu", nn.Linear(2 * H, 2 * Z)), ("var", nn.Linear(2 * H, 2 * Z)),
                ("linear_init_global", nn.Linear(2 * Z + 24, H)), ("grucell_g", nn.GRUCell(2 * Z + 24 + E, H)), ("grucell_g_2", nn.GRUCell(H, H)),
                ("linear_out_g", nn.Linear(H, E))]
    else:
        mods = [("gru_e", nn.GRU(E + 2 if kind == "cvae" else E, H, batch_first=True, bidirectional=True)), ("c_r", nn.Linear(Z, 3)),
                ("c_n", nn.Linear(Z, 3)), ("mu", nn.Linear(2 * H, Z)), ("var", nn.Linear(2 * H, Z))]
        if kind == "fader":
            mods += [("discriminator_r", nn.Linear(Z, 1)), ("discriminator_n", nn.Linear(Z, 1))]
        mods += [("linear_init_global", nn.Linear(Z + 2, H)), ("grucell_g", nn.GRUCell(Z + 2 + E, H)), ("grucell_g_2", nn.GRUCell(H, H)),
                 ("linear_out_g", nn.Linear(H, E))]
    sd = {}
    for name, m in mods:
        for k, v in m.state_dict().items():
            sd["%s.%s" % (name, k)] = v.detach().clone()
    return sd


def trainable_used_keys(sd):
    return [k for k in sd if not k.startswith(("c_r.", "c_n."))]


def encode(sd, kind, x_in):
    gru = "gru." if kind == "single" else "gru_e."
    hf = g._gru_final(x_in, sd, gru, "", False)
    hb = g._gru_final(x_in, sd, gru, "_reverse", True)
    xe = torch.cat([hf, hb], dim=1)
    return xe @ sd["mu.weight"].t() + sd["mu.bias"], torch.exp(xe @ sd["var.weight"].t() + sd["var.bias"])


def forward(sd, kind, d, cond, eps, mask=None, training=True):
    """cond: chroma (single) or [r_density | n_density] (cvae, fader), float32 [B][*]; mask: the scaled dropout keep-masks [B][2]"""
    x = g.convert_to_one_hot(d, E)
    if kind == "cvae":
        x = torch.cat([x, cond.unsqueeze(1).expand(-1, d.shape[1], -1)], dim=-1)
    mu, sg = encode(sd, kind, x)
    z = mu + sg * eps
    fw = dict(mu=mu, sigma=sg, z_lat=z)
    if kind == "fader":
        rz = z                                              # ReverseLayerF forward = identity
        pre = torch.stack([(rz @ sd["discriminator_%s.weight" % e].t() + sd["discriminator_%s.bias" % e]).view(-1) for e in ("r", "n")], dim=1)
        fw["adv_pre"] = pre
        fw["adv_out"] = torch.relu(pre) * (mask if mask is not None else 1.0)
    zc = torch.cat([z, cond], dim=1)
    fw["z"] = zc
    fw["out"] = g.global_decoder(sd, zc, d.shape[1], teacher=d if training else None)
    return fw


def total_loss(sd, kind, batch, eps, mask, step, beta):
    d = torch.as_tensor(batch["d"]).long()
    if kind == "single":
        cond = torch.as_tensor(batch["c"]).float()
    else:
        cond = torch.stack([torch.as_tensor(batch["r_density"]).float(), torch.as_tensor(batch["n_density"]).float()], dim=1)
    fw = forward(sd, kind, d, cond, eps, mask)
    ce = g._nll_mean(fw["out"], d)
    kld = g._kl_normal(fw["mu"], fw["sigma"], torch.zeros_like(fw["mu"]), torch.ones_like(fw["sigma"])).mean()
    beta0 = g.beta_schedule(step, beta)
    if kind == "single":
        ls = []
        for col, a in ((0, batch["r_density"]), (1, batch["n_density"])):
            a = np.asarray(a, np.float64)
            d_attr = torch.from_numpy(np.subtract.outer(a, a)).float()
            zc = fw["z"][:, col]
            ls.append(((torch.tanh(zc.reshape(-1, 1) - zc) - torch.sign(d_attr)) ** 2).mean())
        loss = 5 * ce + beta * kld + ls[0] + ls[1]
        return loss, (loss, ce, ls[0], ls[1]), fw
    if kind == "cvae":
        loss = ce + beta0 * kld
        return loss, (loss, ce), fw
    lam = min(step / 2000 * 1e-4, 1e-4)
    # gradient reversal: the heads' loss reaches the encoder with the opposite sign, the heads' own parameters with the normal one
    z = fw["z_lat"]
    rz = 2.0 * z.detach() - z                            # value z, derivative -1 wrt z: ReverseLayerF
    pre = torch.stack([(rz @ sd["discriminator_%s.weight" % e].t() + sd["discriminator_%s.bias" % e]).view(-1) for e in ("r", "n")], dim=1)
    o = torch.relu(pre) * mask
    la = [lam * ((o[:, a] - cond[:, a]) ** 2).mean() for a in (0, 1)]
    loss = ce + beta0 * kld + la[0] + la[1]
    return loss, (loss, ce, la[0], la[1]), fw


def gradients(sd, kind, batch, eps, mask, step, beta):
    keys = trainable_used_keys(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    p = dict(sd)
    p.update(leaves)
    loss, tup, fw = total_loss(p, kind, batch, eps, mask, step, beta)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    return {k: (gr if gr is not None else torch.zeros_like(sd[k])) for k, gr in zip(keys, grads)}, tuple(float(t.detach()) for t in tup), fw


def draw(kind, B, Z, T, training=True):
    """one forward's draws on torch's global generator: eps, (fader: two dropout masks), T x rand(1) in train mode"""
    eps = torch.randn(B, 2 * Z if kind == "single" else Z)
    mask = None
    if kind == "fader":
        one = torch.ones(B, 1)
        mask = torch.cat([nn.functional.dropout(one, 0.3, training), nn.functional.dropout(one, 0.3, training)], dim=1)
    if training:
        for _ in range(T):
            torch.rand(1)
    return eps, mask
