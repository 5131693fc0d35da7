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


def make_model(hidden, zdim, sd=None, device="cpu", ops=None, seed=1234, arith=None):
    pkg = load_package()
    torch.manual_seed(seed)
    m = pkg.MusicAttrRegGMVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=hidden, z_dims=zdim,
                              n_step=32, n_component=2)
    if sd is not None:
        m.load_state_dict(sd)
    m = m.to(device)
    if ops is not None:
        m._make_ops = lambda dev, _ops=ops: _ops          # test-side injection of the CPU stand-in for the kernel table
    m.train()
    if arith is not None:
        m.set_arith(arith)                                # "f32" / "bf16x6": arithmetic of the deep MFMA products (arith.py)
    return m


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-12, np.abs(b).max()))


def oracle_grads_f64(gold, sd32, sup=False):
    """Gradients of the training loss evaluated by the oracle in float64 ("exact" arithmetic).

    The q(y|x) softmax of gmm_model.py:217 takes log-likelihoods of magnitude ~1e3 whose float32 ulp is ~1e-4; when the
    posterior is not saturated every float32 implementation - the reference included - carries ~1e-3 relative noise in
    the gradients that flow through it.  Parity for those tensors is therefore stated as "no further from the float64
    truth than the reference itself (x4), or 5e-4"."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        sd = {k: v.double() for k, v in sd32.items()}
        grads, tup, _ = orc.gradients(sd, batch_of(gold), torch.from_numpy(gold["eps_r"]).double(),
                                      torch.from_numpy(gold["eps_n"]).double(), 20000, 0.2, is_supervised=sup)
    finally:
        torch.set_default_dtype(old)
    return {k: v.numpy() for k, v in grads.items()}


def grad_tolerances(gold, tag, exact):
    """per-parameter tolerance = max(5e-4, 4 x the reference's own distance from the float64 result)."""
    tol = {}
    for k, ex in exact.items():
        ref = gold["grad_%s/%s" % (tag, k)]
        tol[k] = max(5e-4, 4.0 * relerr(ref, ex))
    return tol


def epoch_loaders(g):
    """the four loaders of tests/golden/epoch.npz, batches laid out as the reference's DataLoaders yield them"""
    def dl(name, n, width):
        out = []
        for i in range(n):
            x = [g["%s%d_%d" % (name, i, j)] for j in range(width)]
            out.append(tuple(torch.from_numpy(np.asarray(t)) if j < width - 2 else np.asarray(t) for j, t in enumerate(x)))
        return out
    return dl("vt", 2, 8), dl("vv", 1, 8), dl("yt", 2, 6), dl("yv", 1, 6)


def check_epoch_run(pkg, m, g, tmp_path, rtol):
    """run pkg.training_phase on the golden loaders and compare every printed number and the saved checkpoint with what the
    reference's own training_phase (trainer_gmm.py:306-467) printed / saved (tests/golden/make_golden_epoch.py)"""
    import re
    tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2)
    vt, vv, yt, yv = epoch_loaders(g)
    lines = []
    save_path = os.path.join(str(tmp_path), "golden.pt")
    torch.manual_seed(4242)
    step = pkg.training_phase(tr, int(g["start_step"]), 2, vt, vv, yt, yv, save_path, name="golden", log=lines.append)
    assert step == int(g["start_step"]) + 8
    ref = [str(l) for l in g["lines"]]
    assert len(lines) == len(ref), (lines, ref)
    num = re.compile(r"-?\d+\.\d+")
    for got, want in zip(lines[:-1], ref[:-1]):
        assert num.sub("#", got) == num.sub("#", want), (got, want)              # same text, same number of fields
        a, b = [float(x) for x in num.findall(got)], [float(x) for x in num.findall(want)]
        np.testing.assert_allclose(a, b, rtol=rtol, atol=2e-4, err_msg=want)     # printed with 4-5 decimals
    assert lines[-1].startswith("Model saved as ") and ref[-1].startswith("Model saved as ")
    saved = torch.load(save_path)
    want_keys = [k[len("wend/"):] for k in g.keys() if k.startswith("wend/")]
    assert sorted(saved.keys()) == sorted(want_keys)
    assert all(v.device.type == "cpu" for v in saved.values())
    stamped = [f for f in os.listdir(str(tmp_path)) if f.startswith("golden_") and f.endswith(".pt")]
    assert len(stamped) == 1
    return saved


def make_vae_model(hidden, zdim, device="cpu", ops=None, seed=1234):
    """seeded MusicAttrRegVAE (model_v2.py:9) from the package, on the test backend `ops` (CPU) or the HIP kernels (device)"""
    pkg = load_package()
    torch.manual_seed(seed)
    m = pkg.MusicAttrRegVAE(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=hidden, z_dims=zdim, n_step=20)
    if ops is not None:
        m._make_ops = lambda dev, _ops=ops: _ops          # test-side injection of the CPU stand-in for the kernel table
    return m.to(device)


def check_vae_against_reference(pkg, m, g, dev, rtol_fw, tol_grad, rtol_tuple, atol_w):
    """drop-in forward, fused gradients, three train() steps and evaluate() of the vanilla-VAE sibling vs tests/golden/vae.npz"""
    t = lambda k, dt=None: torch.from_numpy(g[k]).to(dev) if dt is None else torch.from_numpy(g[k]).to(dev).to(dt)
    d, r, n, c = t("d"), t("r"), t("n"), t("c")
    eps = (t("eps_r"), t("eps_n"))
    sd = m.state_dict()
    for k, v in sd.items():
        np.testing.assert_allclose([float(v.double().sum()), float(v.double().abs().sum())], g["w0sum/" + k], rtol=1e-6, atol=1e-6, err_msg=k)
    assert set(sd) == {k[len("w0sum/"):] for k in g if k.startswith("w0sum/")}
    # drop-in forward: the reference's nested tuple (model_v2.py:165-171)
    (out, r_out, n_out), (dis_r, dis_n), (z_r, z_n) = m(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3), pkg.convert_to_one_hot(n, 16), c, eps=eps)
    got = dict(out=out, r_out=r_out, n_out=n_out, mu_r=dis_r.mean, sigma_r=dis_r.stddev, mu_n=dis_n.mean, sigma_n=dis_n.stddev, z_r=z_r, z_n=z_n)
    for k, v in got.items():
        np.testing.assert_allclose(v.detach().cpu().numpy(), g["fw_" + k], rtol=rtol_fw, atol=rtol_fw, err_msg=k)
    # fused step
    tr = pkg.VAETrainer(m, lr=1e-3, beta=0.1)
    batch = tr.prepare_batch(g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    tup = tr.loss_and_grads(5000, batch, eps)
    np.testing.assert_allclose(tup[:6], g["loss_terms"], rtol=rtol_tuple)
    for k in tr.flat.names:
        ref = g["grad/" + k]
        e = relerr(tr.flat.G[k].cpu().numpy(), ref)
        assert e < tol_grad or np.abs(ref).max() < 1e-6, (k, e)
    assert set(tr.flat.names) == {k[len("grad/"):] for k in g if k.startswith("grad/")}
    np.testing.assert_allclose(tr.grad_norm(), g["gradnorm"][0], rtol=1e-3)
    step = 5000
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = tr.train(step, None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
        assert len(tup) == 6
        np.testing.assert_allclose(tup, g["train_tuples"][it], rtol=rtol_tuple, err_msg="step %d" % it)
    assert step == 5003
    torch.manual_seed(123)
    ev = tr.evaluate(None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    np.testing.assert_allclose(ev, g["eval_tuple"], rtol=rtol_tuple)
    for k, v in m.state_dict().items():
        if k not in NOISE_PARAMS:
            np.testing.assert_allclose(v.cpu().numpy(), g["w3/" + k], rtol=0, atol=atol_w, err_msg=k)


def tokens_match_upto_near_tie(tok, ref_tok, ref_gap, thr=1e-4):
    """greedy tokens are compared per row up to the first position where the reference's own top-2 log-prob gap is below thr
    (a flipped float32 near-tie changes every later token); returns the number of positions compared"""
    tok, ref_tok = np.asarray(tok), np.asarray(ref_tok)
    L = tok.shape[-1]
    n = 0
    for row, rrow, grow in zip(tok.reshape(-1, L), ref_tok.reshape(-1, L), np.asarray(ref_gap).reshape(-1, L)):
        unclear = np.where(grow < thr)[0]
        upto = int(unclear[0]) if len(unclear) else L
        assert np.array_equal(row[:upto], rrow[:upto]), (row[:upto].tolist(), rrow[:upto].tolist())
        n += upto
    return n


def eval_golden(tag):
    return {k[len(tag) + 1:]: v for k, v in load_golden("eval").items() if k.startswith(tag + "/")}


def check_eval_side(pkg, m, g, dev, rtol=2e-5):
    """the product's eval-side callers (evaluators.py, eval-mode forward, fader_sweep) against tests/golden/eval.npz = the reference's
    own evaluator / notebook code run on the same seeds (make_golden_eval.py).  `m` is a freshly seeded model on `dev`."""
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    d, r, n, c = t("d"), t("r"), t("n"), t("c")
    B, T = d.shape
    Z = m.latent_dim
    for k, v in m.state_dict().items():
        np.testing.assert_allclose([float(v.double().sum())], g["w0sum/" + k][:1], rtol=1e-9, atol=1e-9, err_msg=k)
    # D: run_through_gmm (train mode as constructed)
    m.train()
    dl = [(d[i:i + 3], r[i:i + 3], n[i:i + 3], c[i:i + 3], g["r_density"][i:i + 3], g["n_density"][i:i + 3]) for i in range(0, B, 3)]
    torch.manual_seed(5)
    res = pkg.run_through_gmm(m, dl)
    names = ["r_density_lst", "n_density_lst", "r_lst", "n_lst", "a_lst", "r_mean", "n_mean", "z_r_0_lst", "z_r_rest_lst",
             "z_n_0_lst", "z_n_rest_lst", "r_min", "r_max", "n_min", "n_max"]
    for k, v in zip(names, res):
        if k != "a_lst":
            np.testing.assert_allclose(np.asarray(v), g["rt_" + k], rtol=rtol, atol=rtol, err_msg=k)
    # B: evaluator shifts, the reference's call sequence: the first call finds the model in train mode, later ones in eval mode
    m.train()
    ev = {0: pkg.GMMRhythmEvaluator(None), 1: pkg.GMMNoteEvaluator(None)}
    compared = 0
    for k, (which, i, val) in enumerate(g["shift_calls"]):
        i = int(i)
        assert m.training == bool(g["shift%d_training_before" % k][0])
        torch.manual_seed(100 + k)
        out, z0 = ev[int(which)].shift(m, d[i], r[i], n[i], c[i], float(val))
        assert tuple(out.shape) == (1, 100, 342) and not m.training
        np.testing.assert_allclose(z0, g["shift%d_z0" % k][0], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(out[0, 0].cpu().numpy(), g["shift%d_logp0" % k], rtol=1e-4, atol=1e-4)
        compared += tokens_match_upto_near_tie(out.argmax(-1).cpu().numpy(), g["shift%d_tokens" % k], g["shift%d_gap" % k])
        if (g["shift%d_gap" % k] >= 1e-4).all():
            assert np.array_equal(np.asarray(pkg.clean_output(out)), g["shift%d_clean" % k])
    assert compared >= 100
    # the batched form of the same shifts: (sample, value) rows of one decode batch, eps per (sample, value)
    for which in (0, 1):
        calls = [(k, int(i), float(v)) for k, (w, i, v) in enumerate(g["shift_calls"]) if int(w) == which]
        # the reference call k draws [forward eps r, n (+T rand)] then repar r, n: reproduce the repar draws
        eps_r, eps_n = torch.zeros(B, len(calls), Z), torch.zeros(B, len(calls), Z)
        for j, (k, i, v) in enumerate(calls):
            torch.manual_seed(100 + k)
            torch.randn(1, Z), torch.randn(1, Z)
            if bool(g["shift%d_training_before" % k][0]):
                for _ in range(T):
                    torch.rand(1)
            eps_r[i, j], eps_n[i, j] = torch.randn(1, Z)[0], torch.randn(1, Z)[0]
        tk, z0 = pkg.fader_sweep(m, d, c, [v for _, _, v in calls], steps=100, which="rn"[which], eps=(eps_r, eps_n))
        assert tuple(tk.shape) == (B, len(calls), 100) and tuple(z0.shape) == (B, len(calls))
        for j, (k, i, v) in enumerate(calls):
            np.testing.assert_allclose(float(z0[i, j]), g["shift%d_z0" % k][0], rtol=1e-4, atol=1e-5)
            tokens_match_upto_near_tie(tk[i, j].cpu().numpy()[None], g["shift%d_tokens" % k], g["shift%d_gap" % k])
    # A: eval-mode forward: the decoder feeds back its own argmax, no rand(1) draws
    m.eval()
    torch.manual_seed(7)
    (o, r_out, n_out, _, _), dis, z_out, ll_out, qy_out, y_out = m(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3),
                                                                   pkg.convert_to_one_hot(n, 16), c)
    after = torch.rand(1).item()
    torch.manual_seed(7)
    torch.randn(B, Z), torch.randn(B, Z)
    assert after == torch.rand(1).item()
    got = dict(r_out=r_out, n_out=n_out, mu_r=dis[0].mean, sigma_r=dis[0].stddev, z_r=z_out[0], z_n=z_out[1], ll_r=ll_out[0], qy_n=qy_out[1])
    for k, v in got.items():
        np.testing.assert_allclose(v.cpu().numpy(), g["evalfw_" + k], rtol=rtol, atol=rtol, err_msg=k)
    np.testing.assert_allclose(o[:, 0].cpu().numpy(), g["evalfw_logp0"], rtol=1e-4, atol=1e-4)
    assert tuple(o.shape) == (B, T, 342)
    assert tokens_match_upto_near_tie(o.argmax(-1).cpu().numpy(), g["evalfw_tokens"], g["evalfw_gap"]) >= B * T // 2
    assert np.array_equal(y_out[0].cpu().numpy(), g["evalfw_y_r"]) and np.array_equal(y_out[1].cpu().numpy(), g["evalfw_y_n"])
    # C: notebook transfer (cells 11 + 15 / 17), 300 greedy steps, both latents shifted
    for j in range(2):
        i, seed, lmbda, steps = (int(x) for x in g["nb%d_meta" % j])
        torch.manual_seed(seed)
        out, z = pkg.arousal_transfer(m, d[i], c[i], lmbda=lmbda, low_to_high=(j == 0), steps=steps)
        assert tuple(out.shape) == (1, 300, 342)
        np.testing.assert_allclose(z.cpu().numpy(), g["nb%d_z" % j], rtol=1e-4, atol=1e-5)
        assert tokens_match_upto_near_tie(out.argmax(-1).cpu().numpy(), g["nb%d_tokens" % j], g["nb%d_gap" % j]) >= 100
        # and its batched form: mode="shift", which="both", lambda = +-1
        torch.manual_seed(seed)
        eps = (torch.randn(1, Z), torch.randn(1, Z))
        tk, _ = pkg.fader_sweep(m, d[i:i + 1], c[i:i + 1], [lmbda if j == 0 else -lmbda], steps=steps, which="both", mode="shift", eps=eps)
        tokens_match_upto_near_tie(tk[0].cpu().numpy(), g["nb%d_tokens" % j], g["nb%d_gap" % j])
    # train-mode global_decoder(z, steps): teacher forced with self.sample (gmm_model.py:139-142)
    m.train()
    torch.manual_seed(11)
    (o_tf, _, _, _, _), _, z_out, _, _, _ = m(d, r, n, c)
    zc = torch.cat([z_out[0], z_out[1], c], dim=1).detach()
    o2 = m.global_decoder(zc, steps=T)
    assert o2.requires_grad                                      # as in the reference: a train-mode call is part of the autograd graph
    np.testing.assert_allclose(o2.detach().cpu().numpy(), o_tf.detach().cpu().numpy(), rtol=1e-5, atol=1e-5)


# ---- single-encoder siblings (tests/golden/siblings.npz = the reference's model_v2 classes + their own trainers) ----------------------
SIBLINGS = {"single": ("MusicAttrSingleVAE", "SingleVAETrainer"), "cvae": ("MusicAttrCVAE", "CVAETrainer"), "fader": ("MusicAttrFaderNets", "FaderTrainer")}


def sibling_golden(kind):
    return {k[len(kind) + 1:]: v for k, v in load_golden("siblings").items() if k.startswith(kind + "/")}


def make_sibling(kind, hidden, zdim, device="cpu", ops=None, seed=1234):
    pkg = load_package()
    torch.manual_seed(seed)
    m = getattr(pkg, SIBLINGS[kind][0])(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=hidden, z_dims=zdim, n_step=20)
    if ops is not None:
        m._make_ops = lambda dev, _ops=ops: _ops          # test-side injection of the CPU stand-in for the kernel table
    return m.to(device)


def check_sibling(pkg, kind, m, g, dev, rtol_fw=5e-5, tol_grad=5e-4, rtol_tuple=5e-4):
    """drop-in forward (train and eval mode), fused gradients, three train() steps and evaluate() of one sibling against the reference run"""
    H, Z, B, T, Tr = (int(x) for x in g["dims"])
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    d, r, n, c = t("d"), t("r"), t("n"), t("c")
    rd32, nd32 = torch.from_numpy(g["r_density"]).float().unsqueeze(-1).to(dev), torch.from_numpy(g["n_density"]).float().unsqueeze(-1).to(dev)
    for k, v in m.state_dict().items():
        vd = v.double()
        np.testing.assert_allclose([vd.sum().item(), vd.abs().sum().item(), (vd * vd).sum().item()], g["w0sum/" + k], rtol=1e-9, atol=1e-9, err_msg=k)
    assert set(m.state_dict()) == {k[len("w0sum/"):] for k in g if k.startswith("w0sum/")}
    call = (lambda: m(pkg.convert_to_one_hot(d, 342), c)) if kind == "single" else \
           (lambda: m(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3), pkg.convert_to_one_hot(n, 16), c, rd32, nd32))
    # train-mode forward, the reference's nesting, random draws from the seeded global generator
    m.train()
    torch.manual_seed(99)
    res = call()
    if kind == "fader":
        (o, r_out, n_out), dis, z = res
        np.testing.assert_allclose(r_out.detach().cpu().numpy(), g["fw_r_out"], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(n_out.detach().cpu().numpy(), g["fw_n_out"], rtol=1e-4, atol=1e-5)
    else:
        o, dis, z = res
    assert o.requires_grad                                        # a train-mode call is part of the autograd graph, as in the reference
    for k, v in (("out", o), ("mu", dis.mean), ("sigma", dis.stddev), ("z", z)):
        np.testing.assert_allclose(v.detach().cpu().numpy(), g["fw_" + k], rtol=rtol_fw, atol=rtol_fw, err_msg=k)
    # fused gradients
    tr = getattr(pkg, SIBLINGS[kind][1])(m, lr=1e-3, beta=0.2)
    batch = tr.prepare_batch(g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    for step in ((20000, 500) if kind == "fader" else (20000,)):
        torch.manual_seed(99)
        eps = tr.draw_eps(B, T)
        tup = tr.loss_and_grads(step, batch, eps)
        np.testing.assert_allclose(tup, g["loss_terms_%d" % step], rtol=rtol_tuple, atol=1e-9)
        ref_keys = {k[len("grad_%d/" % step):] for k in g if k.startswith("grad_%d/" % step)}
        assert set(tr.flat.names) == ref_keys, (set(tr.flat.names) ^ ref_keys)
        for k in tr.flat.names:
            ref = g["grad_%d/%s" % (step, k)]
            e = relerr(tr.flat.G[k].cpu().numpy(), ref)
            assert e < tol_grad or np.abs(ref).max() < 1e-7, (kind, step, k, e)
        np.testing.assert_allclose(tr.grad_norm(), g["gradnorm_%d" % step][0], rtol=1e-3)
    # the reference's own train() x3 and evaluate()
    step = 19999
    for it in range(3):
        torch.manual_seed(99 + it)
        # the densities as the reference's own loops hand them over: float64 arrays (trainer_singlevae.py:107-120), (B, 1) float32
        # device tensors for the conditional models (trainer_cvae.py:171, trainer_fader.py:180)
        dens = (g["r_density"], g["n_density"]) if kind == "single" or it == 2 else (rd32, nd32)
        step, tup = tr.train(step, None, None, None, g["d"], g["r"], g["n"], g["c"], *dens)
        np.testing.assert_allclose(tup, g["train_tuples"][it], rtol=rtol_tuple, atol=1e-9, err_msg="step %d" % it)
    assert step == 20002
    for k, v in m.state_dict().items():
        vd = v.double()
        np.testing.assert_allclose([vd.abs().sum().item()], g["w3sum/" + k][1:2], rtol=1e-3, err_msg=k)
    torch.manual_seed(123)
    if kind == "cvae":
        ev = tr.evaluate(None, None, None, g["d"], g["r"], g["n"], g["c"], rd32, nd32)
    elif kind == "fader":
        ev = tr.evaluate(step - 1, None, None, None, g["d"], g["r"], g["n"], g["c"], rd32, nd32)
    else:
        ev = tr.evaluate(step - 1, None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    np.testing.assert_allclose(ev, g["eval_tuple"], rtol=rtol_tuple, atol=1e-9)
    # eval-mode forward on fresh weights: greedy decoder
    m2 = make_sibling(kind, H, Z, device=dev, ops=m._make_ops(None) if "_make_ops" in m.__dict__ else None)
    m2.eval()
    d, r, n, c = (x.to(dev) for x in (d, r, n, c))
    torch.manual_seed(7)
    res = (m2(pkg.convert_to_one_hot(d, 342), c) if kind == "single" else
           m2(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3), pkg.convert_to_one_hot(n, 16), c, rd32, nd32))
    o = res[0][0] if kind == "fader" else res[0]
    np.testing.assert_allclose(res[2].cpu().numpy(), g["evalfw_z"], rtol=rtol_fw, atol=rtol_fw)
    np.testing.assert_allclose(o[:, 0].cpu().numpy(), g["evalfw_logp0"], rtol=1e-4, atol=1e-4)
    assert tokens_match_upto_near_tie(o.argmax(-1).cpu().numpy(), g["evalfw_tokens"], g["evalfw_gap"]) >= B * T // 2
    if kind == "fader":
        np.testing.assert_allclose(res[0][1].cpu().numpy(), g["evalfw_r_out"], rtol=1e-4, atol=1e-5)


# ---- GLSR trainer (tests/golden/glsr.npz = trainer_glsr.py's own functions on model_v2.MusicAttrRegVAE) ---------------------------------
def glsr_fixture_weights(g, hidden, zdim):
    """the seeded MusicAttrRegVAE weights with the fixture's output-layer modification (make_golden_glsr.py)"""
    from oracle import vae_oracle as vo
    sd = vo.init_state_dict(hidden, zdim)
    sd["linear_out_g.weight"] = sd["linear_out_g.weight"] * float(g["out_scale"][0])
    sd["linear_out_g.bias"] = sd["linear_out_g.bias"] + torch.from_numpy(g["bias_shift"])
    for k, v in sd.items():
        vd = v.double()
        np.testing.assert_allclose([vd.sum().item(), vd.abs().sum().item(), (vd * vd).sum().item()], g["w0sum/" + k], rtol=1e-6, atol=1e-6, err_msg=k)
    return sd


def check_glsr(pkg, m, g, dev, tol_grad=1e-3, rtol_tuple=5e-4):
    """GLSRTrainer (fused step + four extra teacher-forced decodes + host walk) vs the reference's trainer_glsr.train / evaluate"""
    H, Z, B, T, Tr = (int(x) for x in g["dims"])
    tr = pkg.GLSRTrainer(m, lr=1e-3, beta=0.2)
    batch = tr.prepare_batch(g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    torch.manual_seed(99)
    eps = tr.draw_eps(B, T, step=20000)
    assert len(eps[2]) == 2
    tup = tr.loss_and_grads(20000, batch, eps)[:6]
    np.testing.assert_allclose(tup, g["loss_terms_20000"], rtol=rtol_tuple)
    assert tup[5] > 0.92 and float(g["diag_sep_frac_above"][0]) > 0.2          # the regulariser is not at its constant floor
    ref_keys = {k[len("grad/"):] for k in g if k.startswith("grad/")}
    assert set(tr.flat.names) == ref_keys
    for k in tr.flat.names:
        ref = g["grad/" + k]
        e = relerr(tr.flat.G[k].cpu().numpy(), ref)
        assert e < tol_grad or np.abs(ref).max() < 1e-6, (k, e)
    np.testing.assert_allclose(tr.grad_norm(), g["gradnorm"][0], rtol=2e-3)
    # step <= 20: regulariser off, no extra draws (trainer_glsr.py:289-292)
    sd0 = {k: v.detach().clone() for k, v in m.state_dict().items()}
    torch.manual_seed(50)
    _, t19 = tr.train(19, None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    np.testing.assert_allclose(t19, g["train_tuple_step19"], rtol=rtol_tuple, atol=1e-9)
    # the reference's own train() x3 from the fixture weights, then evaluate()
    m.load_state_dict(sd0)
    tr = pkg.GLSRTrainer(m, lr=1e-3, beta=0.2)
    step = 19999
    for it in range(3):
        torch.manual_seed(99 + it)
        step, tup = tr.train(step, None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
        np.testing.assert_allclose(tup, g["train_tuples"][it], rtol=2e-3 if it else rtol_tuple, err_msg="step %d" % it)
    torch.manual_seed(123)
    ev = tr.evaluate(step - 1, None, None, None, g["d"], g["r"], g["n"], g["c"], g["r_density"], g["n_density"])
    np.testing.assert_allclose(ev, g["eval_tuple"], rtol=2e-3)


# ---- autograd through DIRECT sub-module calls (encode / sub_decoders / global_decoder / approx_qy_x, gmm_model.py:82-218) -----------------
def check_direct_call_autograd(pkg, m, g, dev, tol=5e-4):
    """each direct call in train mode is one autograd node: values and the gradients wrt inputs / parameters against torch autograd of the
    oracle's restatement of the same reference lines, on the `small` fixture's weights and batch"""
    from oracle import gmvae_oracle as orc
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    b = batch_of(g)
    d, r, n = (torch.from_numpy(b[k]) for k in ("d", "r", "n"))
    B, T = d.shape
    Z = m.latent_dim
    m.train()

    def leaves(keys):
        return {k: sd[k].clone().requires_grad_(True) for k in keys}

    def cmp(got, ref, what):
        got, ref = got.detach().cpu().double().numpy(), ref.detach().double().numpy()
        scale = max(1e-6, float(np.abs(ref).max()))
        assert float(np.abs(got - ref).max()) <= tol * scale, (what, float(np.abs(got - ref).max()), scale)

    def zero_grads():
        for p in m.parameters():
            p.grad = None

    # ---- encode ------------------------------------------------------------------------------------------------------
    torch.manual_seed(1)
    w = [torch.randn(B, Z) for _ in range(4)]
    keys = [k for k in sd if k.startswith(("gru_r.", "gru_n.", "mu_r.", "var_r.", "mu_n.", "var_n."))]
    L = leaves(keys)
    p = dict(sd); p.update(L)
    ref = orc.encode(p, orc.convert_to_one_hot(d, 342))
    gref = torch.autograd.grad(sum((o * wi).sum() for o, wi in zip(ref, w)), [L[k] for k in keys])
    zero_grads()
    dis_r, dis_n = m.encode(pkg.convert_to_one_hot(d.to(dev), 342))
    got = (dis_r.mean, dis_r.stddev, dis_n.mean, dis_n.stddev)
    for o, ro, nm in zip(got, ref, ("mu_r", "sigma_r", "mu_n", "sigma_n")):
        cmp(o, ro, "encode " + nm)
    sum((o * wi.to(dev)).sum() for o, wi in zip(got, w)).backward()
    params = dict(m.named_parameters())
    for k, gr in zip(keys, gref):
        cmp(params[k].grad, gr, "encode grad " + k)
    assert params["linear_out_g.weight"].grad is None

    # ---- sub_decoders ----------------------------------------------------------------------------------------------------
    torch.manual_seed(2)
    z_r, z_n = torch.randn(B, Z) * 0.5, torch.randn(B, Z) * 0.5
    keys = [k for k in sd if k.startswith(("gru_d_r.", "gru_d_n.", "linear_out_r.", "linear_out_n.", "linear_init_r.", "linear_init_n."))]
    L = leaves(keys)
    p = dict(sd); p.update(L)
    zr_l, zn_l = z_r.clone().requires_grad_(True), z_n.clone().requires_grad_(True)
    ref_r = orc.sub_decoder(p, "r", orc.convert_to_one_hot(r, 3), zr_l)
    ref_n = orc.sub_decoder(p, "n", orc.convert_to_one_hot(n, 16), zn_l)
    wr, wn = torch.randn_like(ref_r), torch.randn_like(ref_n)
    gref = torch.autograd.grad((ref_r * wr).sum() + (ref_n * wn).sum(), [zr_l, zn_l] + [L[k] for k in keys])
    zero_grads()
    zr_d, zn_d = z_r.to(dev).requires_grad_(True), z_n.to(dev).requires_grad_(True)
    r_out, n_out, _, _ = m.sub_decoders(pkg.convert_to_one_hot(r.to(dev), 3), zr_d, pkg.convert_to_one_hot(n.to(dev), 16), zn_d)
    cmp(r_out, ref_r, "sub_decoders r_out"), cmp(n_out, ref_n, "sub_decoders n_out")
    ((r_out * wr.to(dev)).sum() + (n_out * wn.to(dev)).sum()).backward()
    cmp(zr_d.grad, gref[0], "sub_decoders dz_r"), cmp(zn_d.grad, gref[1], "sub_decoders dz_n")
    for k, gr in zip(keys, gref[2:]):
        if k in ("linear_out_r.bias", "linear_out_n.bias"):       # mathematically zero (time-axis softmax): rounding noise on both sides
            continue
        cmp(params[k].grad, gr, "sub_decoders grad " + k)

    # ---- global_decoder (train mode: teacher forced with self.sample) --------------------------------------------------
    torch.manual_seed(3)
    steps = T - 3
    zc = torch.randn(B, 2 * Z + 24) * 0.5
    keys = [k for k in sd if k.startswith(("linear_out_g.", "grucell_g_2.", "grucell_g.", "linear_init_global."))]
    L = leaves(keys)
    p = dict(sd); p.update(L)
    zc_l = zc.clone().requires_grad_(True)
    teacher = d.long()[:, :steps]
    ref = orc.global_decoder(p, zc_l, steps, teacher=teacher)
    wo = torch.randn_like(ref)
    gref = torch.autograd.grad((ref * wo).sum(), [zc_l] + [L[k] for k in keys])
    zero_grads()
    m.sample = pkg.convert_to_one_hot(d.to(dev), 342)
    zc_d = zc.to(dev).requires_grad_(True)
    state = torch.get_rng_state()
    out = m.global_decoder(zc_d, steps)
    torch.set_rng_state(state)
    for _ in range(steps):
        torch.rand(1)
    probe = torch.rand(1)
    cmp(out, ref, "global_decoder out")
    (out * wo.to(dev)).sum().backward()
    cmp(zc_d.grad, gref[0], "global_decoder dz")
    for k, gr in zip(keys, gref[1:]):
        cmp(params[k].grad, gr, "global_decoder grad " + k)
    assert probe.numel() == 1                                     # (the call drew `steps` x rand(1), as the reference does)

    # ---- approx_qy_x -----------------------------------------------------------------------------------------------------
    torch.manual_seed(4)
    z = (torch.randn(B, Z) * 0.3)
    z_l = z.clone().requires_grad_(True)
    mu_l = sd["mu_r_lookup.weight"].clone().requires_grad_(True)
    ll_ref, qy_ref = orc.approx_qy_x(z_l, mu_l, sd["logvar_r_lookup.weight"])
    w1, w2 = torch.randn_like(ll_ref) * 1e-2, torch.randn_like(qy_ref)
    gref = torch.autograd.grad((ll_ref * w1).sum() + (qy_ref * w2).sum(), [z_l, mu_l])
    zero_grads()
    z_d = z.to(dev).requires_grad_(True)
    ll, qy = m.approx_qy_x(z_d, m.mu_r_lookup, m.logvar_r_lookup, m.n_component)
    cmp(ll, ll_ref, "approx_qy_x ll")
    ((ll * w1.to(dev)).sum() + (qy * w2.to(dev)).sum()).backward()
    cmp(z_d.grad, gref[0], "approx_qy_x dz")
    cmp(m.mu_r_lookup.weight.grad, gref[1], "approx_qy_x dmu_lookup")
    # stale backward is refused; eval mode / no_grad stay forward only
    o1 = m.encode(d.to(dev))[0].mean
    m.encode(d.to(dev))
    try:
        o1.sum().backward()
        raise AssertionError("stale backward accepted")
    except RuntimeError as e:
        assert "must run before the next" in str(e)
    with torch.no_grad():
        assert not m.encode(d.to(dev))[0].mean.requires_grad
    m.eval()
    assert not m.encode(d.to(dev))[0].mean.requires_grad
    m.train()


def check_sibling_autograd(pkg, kind, m, g, dev, tol_grad=5e-4):
    """the reference's own training pattern on the single-encoder drop-ins: forward in train mode, the trainer script's loss written with
    torch ops on the returned tensors (trainer_singlevae.py:87-120, trainer_cvae.py:87-103, trainer_fader.py:87-110), loss.backward() - the
    parameter gradients must be the reference's (siblings.npz grad_20000/*)"""
    from torch.distributions import Normal, kl_divergence
    H, Z, B, T, Tr = (int(x) for x in g["dims"])
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    d, r, n, c = t("d"), t("r"), t("n"), t("c")
    rd32, nd32 = torch.from_numpy(g["r_density"]).float().unsqueeze(-1).to(dev), torch.from_numpy(g["n_density"]).float().unsqueeze(-1).to(dev)
    m.train()
    for p in m.parameters():
        p.grad = None
    torch.manual_seed(99)
    if kind == "single":
        out, dis, z = m(pkg.convert_to_one_hot(d, 342), c)
    else:
        res = m(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3), pkg.convert_to_one_hot(n, 16), c, rd32, nd32)
        (out, r_out, n_out), dis, z = res if kind == "fader" else ((res[0], None, None), res[1], res[2])
    assert out.requires_grad and dis.mean.requires_grad and z.requires_grad
    step, beta = 20000, 0.2
    ce = torch.nn.functional.nll_loss(out.reshape(-1, out.shape[-1]), d.reshape(-1).long())
    kld = kl_divergence(dis, Normal(torch.zeros_like(dis.mean), torch.ones_like(dis.stddev))).mean()
    beta0 = 0.0 if step < 1000 else min((step - 10000) / 10000 * beta, beta)
    if kind == "single":
        regs = []
        for col, a in ((0, g["r_density"]), (1, g["n_density"])):
            da = torch.from_numpy(np.subtract.outer(np.asarray(a, np.float64), np.asarray(a, np.float64))).float().to(dev)
            zc = z[:, col]
            regs.append(((torch.tanh(zc.reshape(-1, 1) - zc) - torch.sign(da)) ** 2).mean())
        loss = 5 * ce + beta * kld + regs[0] + regs[1]
    elif kind == "cvae":
        loss = ce + beta0 * kld
    else:
        lam = min(step / 2000 * 1e-4, 1e-4)
        loss = ce + beta0 * kld + lam * torch.nn.functional.mse_loss(r_out, rd32) + lam * torch.nn.functional.mse_loss(n_out, nd32)
    np.testing.assert_allclose(float(loss.detach()), g["loss_terms_20000"][0], rtol=5e-4)
    loss.backward()
    params = dict(m.named_parameters())
    seen = 0
    for k in [k[len("grad_20000/"):] for k in g if k.startswith("grad_20000/")]:
        ref = g["grad_20000/" + k]
        assert params[k].grad is not None, k
        e = relerr(params[k].grad.cpu().numpy(), ref)
        assert e < tol_grad or np.abs(ref).max() < 1e-7, (kind, k, e)
        seen += 1
    assert seen >= 10
    # an optimiser step on these gradients, then the next forward sees the new weights
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    opt.step()
    m.weights_changed()
    torch.manual_seed(99)
    out2 = (m(pkg.convert_to_one_hot(d, 342), c) if kind == "single" else
            m(pkg.convert_to_one_hot(d, 342), pkg.convert_to_one_hot(r, 3), pkg.convert_to_one_hot(n, 16), c, rd32, nd32))[0]
    out2 = out2[0] if kind == "fader" else out2
    assert float((out2.detach() - out.detach()).abs().max()) > 1e-6


# ---- the five model_config_v2.json trainers: epoch drivers vs the reference's own training_phase (tests/golden/epoch_v2.npz) ---------------
V2_FAMILIES = {"vae": ("MusicAttrRegVAE", "VAETrainer"), "singlevae": ("MusicAttrSingleVAE", "SingleVAETrainer"), "cvae": ("MusicAttrCVAE", "CVAETrainer"),
               "fader": ("MusicAttrFaderNets", "FaderTrainer"), "glsr": ("MusicAttrRegVAE", "GLSRTrainer")}


def make_v2_family(pkg, family, g, device="cpu", ops=None):
    """the seeded model of make_golden_epoch_v2.py for `family` (GLSR: output layer rescaled like the fixture) + its trainer"""
    P = family + "/"
    H, Z, B, T, TR = (int(x) for x in g[P + "dims"])
    torch.manual_seed(1234)
    m = getattr(pkg, V2_FAMILIES[family][0])(roll_dims=342, rhythm_dims=3, note_dims=16, chroma_dims=24, hidden_dims=H, z_dims=Z, n_step=T)
    if family == "glsr":
        with torch.no_grad():
            m.linear_out_g.weight.mul_(float(g[P + "out_scale"][0]))
            m.linear_out_g.bias.add_(torch.from_numpy(g[P + "bias_shift"]))
    if ops is not None:
        m._make_ops = lambda dev, _ops=ops: _ops
    m = m.to(device)
    m.train()
    return m, getattr(pkg, V2_FAMILIES[family][1])(m, lr=1e-3, beta=0.2)


def check_epoch_v2_run(pkg, family, g, tmp_path, device="cpu", ops=None, rtol=5e-4, atol_w=1e-3, noise=()):
    """pkg.training_phase_v2(family) on the golden loaders: same lines (text and numbers) as the reference's training_phase printed, same
    checkpoint key set, weights within atol_w of the saved ones, one time-stamped copy"""
    import re
    P = family + "/"
    m, tr = make_v2_family(pkg, family, g, device, ops)
    dl = lambda name, n: [tuple(torch.from_numpy(np.asarray(g[P + "%s%d_%d" % (name, i, j)])) for j in range(6)) for i in range(n)]
    lines = []
    save_path = os.path.join(str(tmp_path), "golden_%s.pt" % family)
    torch.manual_seed(4242)
    start = int(g[P + "start_step"])
    step = pkg.training_phase_v2(family, tr, start, 2, dl("tr", 2), dl("va", 1), save_path, name="golden_" + family, log=lines.append)
    assert step == start + 4
    ref = [str(l) for l in g[P + "lines"]]
    assert len(lines) == len(ref), (lines, ref)
    num = re.compile(r"-?\d+\.\d+")
    for got, want in zip(lines[:-1], ref[:-1]):
        assert num.sub("#", got) == num.sub("#", want), (got, want)
        a, b = [float(x) for x in num.findall(got)], [float(x) for x in num.findall(want)]
        np.testing.assert_allclose(a, b, rtol=rtol, atol=2e-4, err_msg=want)
    assert lines[-1].startswith("Model saved as ") and ref[-1].startswith("Model saved as ")
    saved = torch.load(save_path)
    want_keys = [k[len(P + "wend/"):] for k in g.keys() if k.startswith(P + "wend/")]
    assert sorted(saved.keys()) == sorted(want_keys)
    assert all(v.device.type == "cpu" for v in saved.values())
    for k, v in saved.items():
        if k not in noise:
            np.testing.assert_allclose(v.numpy(), g[P + "wend/" + k], rtol=0, atol=atol_w, err_msg=k)
    stamped = [f for f in os.listdir(str(tmp_path)) if f.startswith("golden_%s_" % family) and f.endswith(".pt")]
    assert len(stamped) == int(g[P + "stamped"]) == 1
    return m
