"""Eval-side callers of the path, with the reference's names, argument lists and random-number consumption.

  GMMRhythmEvaluator.shift / GMMNoteEvaluator.shift   test_class.py:233-254, :282-303 + test_gmm_v2.py:27-50
  arousal_transfer                                    arousal_transfer.ipynb cells 11 + 15 (low -> high) / 17 (high -> low)
  run_through_gmm                                     test_gmm_v2.py:53-113

RNG contract: the reference draws from torch's global CPU generator (``Normal(0, 1).sample`` == ``torch.randn``,
``dis.rsample()`` == ``torch.randn``); every function here makes the SAME draws in the same order, so
``torch.manual_seed(s); shift(...)`` returns what the reference returns after the same seed (tokens bit-exact up to float32
near-ties; pinned by tests/golden/eval.npz).  The MIDI round trip and the sklearn metrics of ``BaseEvaluator.evaluate``
(test_class.py:79-194) stay out of scope; ``fader_sweep`` (decode.py) is the batched form of the same shift for throughput.
"""
import numpy as np
import torch

from .decode import greedy_decode

EVENT_DIMS = 342


@torch.no_grad()
def _shift(model, d, r, n, c, target_z_value, which, steps):
    dev = model.mu_r.weight.device
    d = torch.as_tensor(d).to(dev)
    if d.dim() == 1 and d.is_floating_point():           # the datasets yield ids as float32 (ptb_v2.py:459-470)
        d = d.long()
    c = torch.as_tensor(c).to(dev).float()
    x = d.unsqueeze(0)                                   # (1, T) ids or (1, T, 342) one-hot
    T = x.shape[1]
    Z = model.latent_dim
    # res = model(d_oh, r_oh, n_oh, c): only `dis` is used below, so the decoders are not run - but the call's random draws are
    # made (2 x randn(1, Z), + T x rand(1) while the model is still in train mode, gmm_model.py:140,230-235)
    model._draw_eps(1, T, "cpu")
    dis_r, dis_n = model.encode(x)
    z_r = dis_r.mean + dis_r.stddev * torch.randn(1, Z).to(dev)          # repar(), test_class.py:53-56,243-244
    z_n = dis_n.mean + dis_n.stddev * torch.randn(1, Z).to(dev)
    tgt = z_r if which == "r" else z_n
    z0 = tgt[:, 0].item()
    tgt[:, 0] = target_z_value                           # "shifting", test_class.py:249 / :298
    model.eval()                                         # and it stays in eval mode, as in the reference (:250)
    z = torch.cat([z_r, z_n, c.view(1, -1)], dim=1)
    return model.global_decoder(z, steps=steps), z0


class GMMRhythmEvaluator:
    """``shift`` of test_class.py:233-254 for the GM-VAE (``handle_*_output`` of test_gmm_v2.py:27-37 folded in)."""
    which = "r"

    def __init__(self, ds=None, epochs=10, num_of_samples=100):
        self.ds, self.epochs, self.num_of_samples = ds, epochs, num_of_samples

    def shift(self, model, d, r, n, c, target_z_value, steps=100):
        """-> (out (1, steps, 342) log-probabilities, value of z[:, 0] before the shift)"""
        return _shift(model, d, r, n, c, target_z_value, self.which, steps)


class GMMNoteEvaluator(GMMRhythmEvaluator):
    """``shift`` of test_class.py:282-303."""
    which = "n"


@torch.no_grad()
def arousal_transfer(model, d, c, lmbda=1.0, low_to_high=True, steps=300):
    """Notebook cells 11 + 15 / 17 for one melody segment: ``model.eval()``, ``z = dis.rsample()`` for r then n, both latents moved by
    ``lmbda * (mu_lookup[1] - mu_lookup[0])`` (opposite sign for high -> low), greedy decode of `steps` tokens.
    -> (out (1, steps, 342) log-probabilities, z (1, 2Z+24))."""
    dev = model.mu_r.weight.device
    model.eval()
    x = torch.as_tensor(d).to(dev)
    if x.dim() == 1 and x.is_floating_point():
        x = x.long()                         # the loaders yield float32 ids (cast like the reference's `.long()`, test_class.py:240)
    x = x.unsqueeze(0)
    Z = model.latent_dim
    dis_r, dis_n = model.encode(x)
    z_r = dis_r.mean + dis_r.stddev * torch.randn(1, Z).to(dev)
    z_n = dis_n.mean + dis_n.stddev * torch.randn(1, Z).to(dev)
    sgn = 1.0 if low_to_high else -1.0
    kidx = torch.arange(0, 2, device=dev)
    mu_r, mu_n = model.mu_r_lookup(kidx), model.mu_n_lookup(kidx)        # cell 11: the two component means of each space
    z_r = z_r + lmbda * sgn * (mu_r[1] - mu_r[0])
    z_n = z_n + lmbda * sgn * (mu_n[1] - mu_n[0])
    z = torch.cat([z_r, z_n, torch.as_tensor(c).to(dev).float().view(1, -1)], dim=1)
    return model.global_decoder(z, steps=steps), z


@torch.no_grad()
def run_through_gmm(model, dl):
    """test_gmm_v2.py:53-113: one forward per batch ``(d, r, n, c, r_density, n_density)``, collecting z and the posterior means;
    returns the reference's 15-tuple.  Forward only: nothing is saved for a backward (the reference keeps autograd graphs alive
    for nothing here)."""
    dev = model.mu_r.weight.device
    acc = {k: [] for k in ("r", "n", "rd", "nd", "zr", "zn", "mr", "mn")}
    for d, r, n, c, r_density, n_density in dl:
        d, r, n = (torch.as_tensor(x).to(dev).long() for x in (d, r, n))
        c = torch.as_tensor(c).to(dev).float()
        _, dis, z_out, _, _, _ = model(d, r, n, c)
        acc["r"].append(r.cpu()), acc["n"].append(n.cpu())
        acc["rd"].append(torch.as_tensor(r_density).float()), acc["nd"].append(torch.as_tensor(n_density).float())
        acc["zr"].append(z_out[0].cpu()), acc["zn"].append(z_out[1].cpu())
        acc["mr"].append(dis[0].mean.cpu()), acc["mn"].append(dis[1].mean.cpu())
    cat = {k: torch.cat(v, dim=0).numpy() for k, v in acc.items()}
    zr, zn = cat["zr"], cat["zn"]
    return (cat["rd"], cat["nd"], cat["r"], cat["n"], [], cat["mr"], cat["mn"], zr[:, 0], zr[:, 1:], zn[:, 0], zn[:, 1:],
            np.amin(zr[:, 0]), np.amax(zr[:, 0]), np.amin(zn[:, 0]), np.amax(zn[:, 0]))
