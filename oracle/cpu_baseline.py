"""CPU baseline for bench.py's `cpu_baseline` leg  --  TEST / MEASUREMENT INFRASTRUCTURE, NOT PRODUCT.

What the reference executes on its CPU path, restated with the same torch operators it uses (nn.GRU, nn.GRUCell,
nn.Linear on DENSE one-hot inputs, trainer_gmm.py:296-303; autograd; clip_grad_norm_; Adam), all host threads.
The reference's own files cannot travel to the GPU box, so this is the timed stand-in; tests/test_oracle_golden.py
checks that it reproduces the golden losses of the imported reference.
"""
import time

import numpy as np
import torch
from torch import nn
from torch.nn import functional as F

from . import gmvae_oracle as orc


class TorchModuleModel(nn.Module):
    """Same topology / operators as gmm_model.py:33-149 (only the modules that forward uses)."""

    def __init__(self, H, Z, K):
        super().__init__()
        E, R, N = orc.E, orc.R, orc.N
        self.gru_r = nn.GRU(E, H, batch_first=True, bidirectional=True)
        self.gru_n = nn.GRU(E, H, batch_first=True, bidirectional=True)
        self.gru_d_r = nn.GRU(Z + R, H, batch_first=True)
        self.gru_d_n = nn.GRU(Z + N, H, batch_first=True)
        self.mu_r, self.var_r = nn.Linear(2 * H, Z), nn.Linear(2 * H, Z)
        self.mu_n, self.var_n = nn.Linear(2 * H, Z), nn.Linear(2 * H, Z)
        self.linear_init_global = nn.Linear(2 * Z + 24, H)
        self.grucell_g = nn.GRUCell(2 * Z + 24 + E, H)
        self.grucell_g_2 = nn.GRUCell(H, H)
        self.linear_init_r, self.linear_init_n = nn.Linear(Z, H), nn.Linear(Z, H)
        self.linear_out_r, self.linear_out_n, self.linear_out_g = nn.Linear(H, R), nn.Linear(H, N), nn.Linear(H, E)
        self.mu_r_lookup, self.mu_n_lookup = nn.Embedding(K, Z), nn.Embedding(K, Z)
        self.logvar_r_lookup, self.logvar_n_lookup = nn.Embedding(K, Z), nn.Embedding(K, Z)
        self.logvar_r_lookup.weight.requires_grad = False
        self.logvar_n_lookup.weight.requires_grad = False

    def forward(self, x, r_oh, n_oh, c, eps_r, eps_n):
        B, T, _ = x.shape
        res = {}
        for e, gru, mu, var, eps in (("r", self.gru_r, self.mu_r, self.var_r, eps_r), ("n", self.gru_n, self.mu_n, self.var_n, eps_n)):
            h = gru(x)[-1].transpose(0, 1).contiguous().view(B, -1)
            res["mu_" + e], res["sigma_" + e] = mu(h), var(h).exp()
            res["z_" + e] = res["mu_" + e] + res["sigma_" + e] * eps
            lk = getattr(self, "mu_%s_lookup" % e).weight
            lv = getattr(self, "logvar_%s_lookup" % e).weight
            res["ll_" + e], res["qy_" + e] = orc.approx_qy_x(res["z_" + e], lk, lv)
        for e, oh, gru, li, lo in (("r", r_oh, self.gru_d_r, self.linear_init_r, self.linear_out_r),
                                   ("n", n_oh, self.gru_d_n, self.linear_init_n, self.linear_out_n)):
            z = res["z_" + e]
            inp = torch.cat([oh, torch.stack([z] * oh.shape[1], dim=1)], dim=-1)
            out = gru(inp, li(z).unsqueeze(0))[0]
            res[e + "_out"] = F.log_softmax(lo(out), 1)
        z = torch.cat([res["z_r"], res["z_n"], c], dim=1)
        out = torch.zeros(B, orc.E)
        out[:, -1] = 1.0
        hx0, hx1, xs = self.linear_init_global(z), None, []
        for i in range(T):
            hx0 = self.grucell_g(torch.cat([out, z], 1), hx0)
            if i == 0:
                hx1 = hx0
            hx1 = self.grucell_g_2(hx0, hx1)
            o = F.log_softmax(self.linear_out_g(hx1), 1)
            xs.append(o)
            out = x[:, i, :]
        res["out"] = torch.stack(xs, 1)
        return res


def build(sd, H, Z, K=2):
    m = TorchModuleModel(H, Z, K)
    own = m.state_dict()
    m.load_state_dict({k: sd[k] for k in own})
    return m


def train_step(model, opt, batch, eps_r, eps_n, step, beta=0.2):
    """trainer_gmm.py:220-258 with the torch operators of the reference; returns the 8 numbers."""
    d, r, n = (torch.as_tensor(batch[k]).long() for k in ("d", "r", "n"))
    c = torch.as_tensor(batch["c"]).float()
    x, r_oh, n_oh = orc.convert_to_one_hot(d, orc.E), orc.convert_to_one_hot(r, orc.R), orc.convert_to_one_hot(n, orc.N)
    opt.zero_grad()
    fw = model(x, r_oh, n_oh, c, eps_r, eps_n)
    sd = dict(model.named_parameters())
    ls = orc.loss_function(sd, fw, d, r, n, step, beta=beta)
    l_r, l_n = orc.latent_regularized_loss(fw["z_r"], fw["z_n"], batch["r_density"], batch["n_density"])
    loss = ls[0] + l_r + l_n
    loss.backward()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
    opt.step()
    return tuple(float(v.detach()) for v in (loss, ls[1], ls[2], ls[3], l_r, l_n, ls[4] + ls[5], ls[6] + ls[7]))


def host_threads():
    """usable host cores: the scheduler affinity of this process (os.cpu_count() over-reports inside containers)."""
    import os
    try:
        return max(1, len(os.sched_getaffinity(0)))
    except AttributeError:
        return os.cpu_count() or 1


MAX_THREADS = 64


def pick_threads(probe=None, threads=None):
    """SURVEY.md 8d asks for all host cores with the count printed.  On the pool's 256-core hosts torch's intra-op pool is pathological with
    256 threads - measured in round 4: ONE 32-row x 256-step train step took 2814 s with 256 threads against 7.0 s with 64 - so the
    baseline runs with min(usable cores, 64) threads (`cores`) and reports the host's core counts next to it (`host_cpus`, `usable_cpus`);
    no all-core probe is made (it would cost the bench run most of an hour).  -> (threads, note)"""
    if threads:
        torch.set_num_threads(threads)
        return threads, "threads fixed by the caller"
    allc = host_threads()
    n = min(allc, MAX_THREADS)
    torch.set_num_threads(n)
    if n == allc:
        return n, "all %d usable cores" % allc
    return n, ("%d of %d usable cores (torch's intra-op pool is pathological beyond that on this host class: 2814 s against 7.0 s for one "
               "32-row step with 256 / 64 threads, measured round 4)" % (n, allc))


def time_baseline(H, Z, B, T, Tr, seed=0, threads=None, max_step_s=120.0, fallback_B=32):
    """tokens/s of the CPU path: the benchmark shape itself, 1 warm-up step + 2 timed steps; only when the warm-up
    step takes longer than `max_step_s` the sample falls back to `fallback_B` rows of the same T-step sequences (again 1 + 2 steps)."""
    import os
    from importlib import import_module
    sd = orc.init_state_dict(H, Z)
    model = build(sd, H, Z)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    synth = import_module("music_fader_nets_amd.synth")
    warm = synth.synth_batch(np.random.RandomState(1), 4, 16, 4)
    train_step(model, opt, warm, torch.randn(4, Z), torch.randn(4, Z), 20000)          # thread-pool / allocator warm-up

    threads, tnote = pick_threads(None, threads)

    def run(b_rows, timed):
        b = synth.synth_batch(np.random.RandomState(seed), b_rows, T, Tr)
        torch.manual_seed(99)
        eps_r, eps_n = torch.randn(b_rows, Z), torch.randn(b_rows, Z)
        t0 = time.perf_counter()
        train_step(model, opt, b, eps_r, eps_n, 20000)
        t_warm = time.perf_counter() - t0
        if not timed(t_warm):
            return t_warm, None
        t0 = time.perf_counter()
        for s in range(2):
            train_step(model, opt, b, eps_r, eps_n, 20001 + s)
        return t_warm, (time.perf_counter() - t0) / 2

    b_used = B
    t_warm, dt = run(B, lambda tw: tw <= max_step_s)
    note = ""
    if dt is None:
        note = " (B=%d warm-up step took %.0f s > %.0f s: fell back)" % (B, t_warm, max_step_s)
        b_used = fallback_B
        t_warm, dt = run(fallback_B, lambda tw: True)
    return dict(value=b_used * T / dt, unit="event-tokens/s", cores=threads, host_cpus=os.cpu_count(), usable_cpus=host_threads(), kind="port",
                sample="B=%d x T=%d (Tr=%d): 1 warm-up step (%.1f s) + 2 timed steps of the dense-one-hot torch.nn train step, "
                       "%.1f s/step%s; threads: %s" % (b_used, T, Tr, t_warm, dt, note, tnote))


def time_decode_baseline(H, Z, rows, steps, seed=0, threads=None):
    """tokens/s of the reference's eval-mode ``global_decoder`` (gmm_model.py:119-149, argmax feedback) on the host cores:
    `rows` sequences x `steps` greedy steps through the torch.nn restatement (dense one-hot inputs, as the reference executes).  Every
    greedy step costs the same (no state grows with the step index), so tokens/s of `steps` steps is tokens/s of any longer decode."""
    import os
    sd = orc.init_state_dict(H, Z)
    torch.manual_seed(seed)
    z = torch.randn(rows, 2 * Z + 24)

    threads, tnote = pick_threads(None, threads)
    orc.greedy_decode(sd, z[:2], 2)                                                      # warm-up
    t0 = time.perf_counter()
    orc.greedy_decode(sd, z, steps)
    dt = time.perf_counter() - t0
    return dict(value=rows * steps / dt, unit="event-tokens/s", cores=threads, host_cpus=os.cpu_count(), usable_cpus=host_threads(), kind="port",
                sample="%d sequences x %d greedy steps of the eval-mode global_decoder restatement, %.1f s (the GPU leg decodes the same %d "
                       "rows for 300 steps; every step costs the same, so the rate carries over); threads: %s" % (rows, steps, dt, rows, tnote))
