"""CPU oracle for the vanilla-VAE sibling MusicAttrRegVAE  --  TEST INFRASTRUCTURE, NOT PRODUCT.

Only ``tests/`` may import this file.  From-scratch fp32 restatement of

    model          model_v2.py:9-171   (encoder :81-97, sub_decoders :99-116, global_decoder :118-143, forward :145-171)
    losses / step  trainer.py:87-162   (loss_function, latent regulariser, train: clip_grad_norm_(.,1) + Adam)

built on the blocks of ``gmvae_oracle`` that the two models share verbatim (GRU recurrences, time-axis log-softmax heads,
teacher-forced decoder).  Parity pin: ``tests/golden/vae.npz``, produced by importing the reference itself
(``tests/golden/make_golden_vae.py``) - see ``tests/test_oracle_golden.py``.

Quirk restated on purpose: ``loss_function`` reads the module-level ``step`` (trainer.py:57,93) which stays 0, so beta0 == 0 and
the KL-to-N(0,1) term (:106-111) never enters the loss.
"""
import torch

from . import gmvae_oracle as g

E, R, N, C = g.E, g.R, g.N, g.C
UNUSED_PREFIXES = g.UNUSED_PREFIXES


def init_state_dict(hidden, zdim, seed=1234):
    """``torch.manual_seed(seed); MusicAttrRegVAE(...)``: the torch constructors in the order of model_v2.py:26-60."""
    from torch import nn
    torch.manual_seed(seed)
    H, Z = hidden, zdim
    mods = [
        ("gru_r", nn.GRU(E, H, batch_first=True, bidirectional=True)),
        ("gru_n", nn.GRU(E, H, batch_first=True, bidirectional=True)),
        ("gru_c", nn.GRU(E, H, batch_first=True, bidirectional=True)),
        ("gru_d_r", nn.GRU(Z + R, H, batch_first=True)),
        ("gru_d_n", nn.GRU(Z + N, H, batch_first=True)),
        ("gru_d_c", nn.GRU(Z + C, H, batch_first=True)),
        ("c_r", nn.Linear(Z, 3)), ("c_n", nn.Linear(Z, 3)),
        ("mu_r", nn.Linear(2 * H, Z)), ("var_r", nn.Linear(2 * H, Z)),
        ("mu_n", nn.Linear(2 * H, Z)), ("var_n", nn.Linear(2 * H, Z)),
        ("mu_c", nn.Linear(2 * H, Z)), ("var_c", nn.Linear(2 * H, Z)),
        ("linear_init_global", nn.Linear(2 * Z + 24, H)),
        ("grucell_g", nn.GRUCell(2 * Z + 24 + E, H)),
        ("grucell_g_2", nn.GRUCell(H, H)),
        ("linear_init_r", nn.Linear(Z, H)), ("linear_init_n", nn.Linear(Z, H)), ("linear_init_c", nn.Linear(Z, H)),
        ("linear_out_r", nn.Linear(H, R)), ("linear_out_n", nn.Linear(H, N)),
        ("linear_out_c", nn.Linear(Z, C)), ("linear_out_g", nn.Linear(H, E)),
    ]
    sd = {}
    for name, m in mods:
        for k, v in m.state_dict().items():
            sd["%s.%s" % (name, k)] = v.detach().clone()
    return sd


def trainable_used_keys(sd):
    return [k for k in sd if not k.startswith(UNUSED_PREFIXES)]


def forward(sd, d, r, n, c, eps_r, eps_n):
    """model_v2.py:145-171 in train mode (int tokens in, as in gmvae_oracle.forward)."""
    mu_r, sg_r, mu_n, sg_n = g.encode(sd, g.convert_to_one_hot(d, E))
    z_r = mu_r + sg_r * eps_r
    z_n = mu_n + sg_n * eps_n
    r_out = g.sub_decoder(sd, "r", g.convert_to_one_hot(r, R), z_r)
    n_out = g.sub_decoder(sd, "n", g.convert_to_one_hot(n, N), z_n)
    out = g.global_decoder(sd, torch.cat([z_r, z_n, c], dim=1), d.shape[1], teacher=d)
    return dict(out=out, r_out=r_out, n_out=n_out, mu_r=mu_r, sigma_r=sg_r, mu_n=mu_n, sigma_n=sg_n, z_r=z_r, z_n=z_n)


def loss_function(fw, d, r, n, beta=0.1, global_step=0):
    """trainer.py:87-114 -> (loss, CE_X, CE_R, CE_N); ``global_step`` is the module-level ``step`` the reference reads (always 0)."""
    beta0 = 0.0 if global_step < 1000 else min((global_step - 10000) / 10000 * beta, beta)
    ce_x, ce_r, ce_n = g._nll_mean(fw["out"], d), g._nll_mean(fw["r_out"], r), g._nll_mean(fw["n_out"], n)
    kld = 0.0
    for e in ("r", "n"):
        mu, sg = fw["mu_" + e], fw["sigma_" + e]
        kld = kld + g._kl_normal(mu, sg, torch.zeros_like(mu), torch.ones_like(sg)).mean()
    return 5 * ce_x + ce_r + ce_n + beta0 * kld, ce_x, ce_r, ce_n


def total_loss(sd, batch, eps_r, eps_n, beta=0.1):
    d, r, n = (torch.as_tensor(batch[k]).long() for k in ("d", "r", "n"))
    fw = forward(sd, d, r, n, torch.as_tensor(batch["c"]).float(), eps_r, eps_n)
    ls = loss_function(fw, d, r, n, beta)
    l_r, l_n = g.latent_regularized_loss(fw["z_r"], fw["z_n"], batch["r_density"], batch["n_density"])
    loss = ls[0] + l_r + l_n
    return loss, (loss, ls[1], ls[2], ls[3], l_r, l_n), fw


def gradients(sd, batch, eps_r, eps_n, beta=0.1):
    keys = trainable_used_keys(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    p = dict(sd)
    p.update(leaves)
    loss, tup, fw = total_loss(p, batch, eps_r, eps_n, beta)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    return {k: (gr if gr is not None else torch.zeros_like(sd[k])) for k, gr in zip(keys, grads)}, tup, fw


# ----------------------------------------------------------------------------------------------------------------------------------
# GLSR variant of the step (trainer_glsr.py:87-321): same model, the loss uses the PASSED step for beta0, and after step 20 the
# finite-difference regulariser of :118-264 replaces the pairwise one.  Pinned by tests/golden/glsr.npz (make_golden_glsr.py).
# ----------------------------------------------------------------------------------------------------------------------------------
NOTES, SEPS, GLSR_EPS, GLSR_STEPS = (2, 90), (180, 278), 1e-2, 100


def draw_glsr(B, Z, T, step):
    """the draws of one trainer_glsr.train() call on torch's global generator: forward (randn x2, T x rand(1)); if step > 20, per
    latent rand(B) for the deltas followed by the 2 x 100 rand(1) of the two train-mode decodes"""
    eps_r, eps_n = torch.randn(B, Z), torch.randn(B, Z)
    for _ in range(T):
        torch.rand(1)
    deltas = []
    if step > 20:
        for _ in range(2):
            deltas.append((1 + torch.rand(B)) * GLSR_EPS)
            for _ in range(2 * GLSR_STEPS):
                torch.rand(1)
    return eps_r, eps_n, deltas


def _rhythm_density(probs):
    """trainer_glsr.py:139-165 incl. its use of ``played_notes[0]`` (sample 0) for every row; probs [B][steps][E]"""
    played = probs[:, :, NOTES[0]:NOTES[1]].sum(-1)
    sep = probs[:, :, SEPS[0]:SEPS[1]].sum(-1)
    res = []
    for b in range(probs.shape[0]):
        total, cur, started = 0, 0, False
        for i in range(probs.shape[1]):
            if sep[b, i].item() < 0.9:
                cur = cur + played[0, i]
                started = True
            else:
                if not started or float(cur) == 0:
                    continue
                total = total + (cur / cur if float(cur) > 1e-2 else cur)
                cur, started = 0, False
        r = total / sep[b].sum()
        res.append(r if float(r) != 0.0 else torch.zeros(()))
    return torch.stack(res)


def glsr_terms(sd, z_r, z_n, c, d, deltas):
    """-> (l_r, l_n) of trainer_glsr.py:167-264 (differentiable wrt sd / z)"""
    import math
    teacher = d[:, :GLSR_STEPS]
    out = []
    for attr, dl in enumerate(deltas):
        vals = []
        for sign in (1.0, -1.0):
            zr, zn = z_r.clone(), z_n.clone()
            (zr if attr == 0 else zn)[:, 0] += sign * dl
            probs = g.global_decoder(sd, torch.cat([zr, zn, c], dim=1), GLSR_STEPS, teacher=teacher).exp()
            vals.append(_rhythm_density(probs) if attr == 0 else probs[:, :, NOTES[0]:NOTES[1]].sum(-1).sum(1))
        grad_attr = (vals[0] - vals[1]) / (2 * dl)
        out.append((0.5 * grad_attr ** 2 + 0.5 * math.log(2 * math.pi)).mean())
    return out[0], out[1]


def glsr_total_loss(sd, batch, eps_r, eps_n, deltas, step, beta):
    d, r, n = (torch.as_tensor(batch[k]).long() for k in ("d", "r", "n"))
    c = torch.as_tensor(batch["c"]).float()
    fw = forward(sd, d, r, n, c, eps_r, eps_n)
    ls = loss_function(fw, d, r, n, beta, global_step=step)
    zero = torch.zeros(())
    l_r, l_n = glsr_terms(sd, fw["z_r"], fw["z_n"], c, d, deltas) if step > 20 else (zero, zero)
    loss = ls[0] + l_r + l_n
    return loss, (loss, ls[1], ls[2], ls[3], l_r, l_n), fw


def glsr_gradients(sd, batch, eps_r, eps_n, deltas, step, beta):
    keys = trainable_used_keys(sd)
    leaves = {k: sd[k].clone().requires_grad_(True) for k in keys}
    p = dict(sd)
    p.update(leaves)
    loss, tup, fw = glsr_total_loss(p, batch, eps_r, eps_n, deltas, step, beta)
    grads = torch.autograd.grad(loss, [leaves[k] for k in keys], allow_unused=True)
    return {k: (gr if gr is not None else torch.zeros_like(sd[k])) for k, gr in zip(keys, grads)}, tuple(float(t.detach()) for t in tup), fw
