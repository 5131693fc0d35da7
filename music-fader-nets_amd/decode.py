"""Eval-mode global decoder: greedy autoregressive decode (gmm_model.py:119-149 with model.eval(), i.e.
``out = self._sampling(out)`` feedback, :147-148) and the fader-shift drivers of test_class.py:233-254,
:282-303 / arousal_transfer.ipynb cells 11+15, batched over many samples x fader values.

Per step: layer-1 cell (token row gather + recurrent MFMA GEMM + gates), layer-2 cell, 512->342 output
GEMM, log_softmax + first-index argmax written straight into the token matrix the next step reads.
"""
import numpy as np
import torch

from .engine import E_VOCAB, LOGIT_LD


@torch.no_grad()
def greedy_decode(model, z, steps, want_logp=True, use_graph=None):
    """z (Bi, 2Z+24) -> (log-probs (Bi, steps, 342) or None, tokens (Bi, steps) int32).

    On the GPU the whole decode (steps x {layer-1 cell, W_ih2 projection, layer-2 cell, output GEMM, log_softmax+argmax}) is
    captured once per (Bi, steps) into a hipGraph and replayed: the loop is launch-latency bound (7 small kernels per token)."""
    eng = model.engine()
    z = z.float().contiguous()
    if _single_launch_ok(eng, z):
        return _decode_single_launch(model, eng, z, steps, want_logp)
    if use_graph is None:
        use_graph = z.is_cuda
    if not use_graph:
        return _decode_body(eng, z, steps, want_logp, None, None)
    cache = eng.__dict__.setdefault("_decode_graphs", {})
    key = (z.shape[0], steps, bool(want_logp), model._version)
    ent = cache.get(key)
    if ent is None:
        for k in [k for k in cache if k[3] != model._version]:     # weights changed: captured weight images are stale
            del cache[k]
        zs = z.clone()
        tokens = torch.zeros(z.shape[0], steps, dtype=torch.int32, device=z.device)
        logp = torch.empty(z.shape[0], steps, E_VOCAB, device=z.device) if want_logp else None
        _decode_body(eng, zs, min(steps, 2), want_logp, logp, tokens)                # warm-up: allocates every buffer
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            _decode_body(eng, zs, steps, want_logp, logp, tokens)
        ent = cache[key] = (g, zs, logp, tokens)
    g, zs, logp, tokens = ent
    zs.copy_(z)
    g.replay()
    return (None if logp is None else logp.clone()), tokens.clone()


def _single_launch_ok(eng, z):
    """small batches decode as ONE launch (fn_decode_greedy: weight slices resident in LDS, activations handed over through L2)"""
    import os
    return (hasattr(eng.ops, "decode_greedy") and z.is_cuda and z.shape[0] <= 32 and eng.H <= 512 and
            os.environ.get("FN_DECODE_PERSIST", "1") == "1")


def _decode_single_launch(model, eng, z, steps, want_logp):
    ops, P, H = eng.ops, eng.p, eng.H
    Bi = z.shape[0]
    packs = eng.__dict__.setdefault("_decode_packs", {})
    if packs.get("version") != model._version:             # operand images of the two matrices the training step never packs
        packs.clear()
        for key, name in (("ih2", "grucell_g_2.weight_ih"), ("out", "linear_out_g.weight")):
            w = P[name]
            packs[key] = torch.zeros(ops.frag_floats(w.shape[0], w.shape[1]), device=z.device)
            ops.frag_pack(w, packs[key])
        packs["version"] = model._version
    h0g = eng.buf("dec_h0g", (Bi, H))
    ops.gemm(z, P["linear_init_global.weight"], h0g, bias=P["linear_init_global.bias"])
    rbg = eng.buf("dec_rbg", (Bi, 3 * H))
    ops.gemm(z, P["grucell_g.weight_ih"][:, E_VOCAB:], rbg)
    tokens = torch.zeros(Bi, steps, dtype=torch.int32, device=z.device)
    logp = torch.empty(Bi, steps, E_VOCAB, device=z.device) if want_logp else None
    ops.decode_greedy(Bi, steps, H, E_VOCAB, E_VOCAB - 1, eng.whh_f["g"], P["grucell_g.bias_hh"], P["grucell_g.bias_ih"], eng.tab["g"], rbg, h0g,
                      packs["ih2"], P["grucell_g_2.bias_ih"], eng.whh_f["g2"], P["grucell_g_2.bias_hh"], packs["out"], P["linear_out_g.bias"],
                      tokens, logp)
    return logp, tokens


def _decode_body(eng, z, steps, want_logp, logp, tokens):
    ops, P, H = eng.ops, eng.p, eng.H
    Bi = z.shape[0]
    dev = z.device
    if tokens is None:
        tokens = torch.zeros(Bi, steps, dtype=torch.int32, device=dev)
    if logp is None and want_logp:
        logp = torch.empty(Bi, steps, E_VOCAB, device=dev)
    hx0 = [eng.buf("dec_hx0_a", (1, Bi, H)), eng.buf("dec_hx0_b", (1, Bi, H))]
    hx1 = [eng.buf("dec_hx1_a", (1, Bi, H)), eng.buf("dec_hx1_b", (1, Bi, H))]
    h0g = eng.buf("dec_h0g", (Bi, H))
    ops.gemm(z, P["linear_init_global.weight"], h0g, bias=P["linear_init_global.bias"])
    rbg = eng.buf("dec_rbg", (Bi, 3 * H))
    ops.gemm(z, P["grucell_g.weight_ih"][:, E_VOCAB:], rbg)
    gx2 = eng.buf("dec_gx2", (1, Bi, 3 * H))
    logits = eng.buf("dec_logits", (Bi, LOGIT_LD))
    # every cell also leaves its new state in the MFMA operand layout, which the next cell call takes as h0_frag: no packing launches
    nf = ops.frag_floats(Bi, H)
    hf0 = [eng.buf("dec_hf0_a", (nf,)), eng.buf("dec_hf0_b", (nf,))]
    hf1 = [eng.buf("dec_hf1_a", (nf,)), eng.buf("dec_hf1_b", (nf,))]
    for i in range(steps):
        cur, prv = i & 1, (i & 1) ^ 1
        ops.gru_seq_fwd([dict(B=Bi, T=1, H=H, w_hh_frag=eng.whh_f["g"], b_hh=P["grucell_g.bias_hh"], b_ih=P["grucell_g.bias_ih"],
                              h0=h0g if i == 0 else hx0[prv][0], h0_frag=None if i == 0 else hf0[prv], h_last_frag=hf0[cur],
                              gx_table=eng.tab["g"], idx=tokens, idx_shift=i - 1,
                              start_token=E_VOCAB - 1, gx_rowbias=rbg, h_all=hx0[cur])])
        ops.gemm(hx0[cur][0], P["grucell_g_2.weight_ih"], gx2[0], bias=P["grucell_g_2.bias_ih"])
        ops.gru_seq_fwd([dict(B=Bi, T=1, H=H, w_hh_frag=eng.whh_f["g2"], b_hh=P["grucell_g_2.bias_hh"],
                              h0=hx0[cur][0] if i == 0 else hx1[prv][0], h0_frag=hf0[cur] if i == 0 else hf1[prv], h_last_frag=hf1[cur],
                              gx_dense=gx2, h_all=hx1[cur])])
        ops.gemm(hx1[cur][0], P["linear_out_g.weight"], logits[:, :E_VOCAB], bias=P["linear_out_g.bias"])
        ops.vocab_argmax(logits, E_VOCAB, logp[:, i, :] if want_logp else None, tokens[:, i])
    return logp, tokens


def clean_output(out):
    """test_class.py:44-50: argmax over the vocabulary -> trim zeros at both ends -> cut at the first EOS (token 1).
    Accepts log-probs (1, steps, 342) or an int token row."""
    if torch.is_tensor(out) and out.is_floating_point():
        recon = torch.argmax(out, dim=-1).cpu().numpy().squeeze()
    else:
        recon = np.asarray(out.cpu() if torch.is_tensor(out) else out).squeeze()
    recon = np.trim_zeros(np.atleast_1d(recon))
    if 1 in recon:
        last = np.argwhere(recon == 1)[0][0]
        recon[recon == 1] = 0
        recon = recon[:last]
    return recon


@torch.no_grad()
def fader_sweep(model, x, chroma, values, steps=100, which="r", eps=None, mode="set"):
    """Batched RhythmEvaluator.shift / NoteEvaluator.shift (test_class.py:233-254, :282-303) and the notebook's
    lambda*shift-vector transfer (cell 15).

    x (n, T) token ids or (n, T, 342) one-hot; chroma (n, 24); values: fader values.
      mode="set":   z_which[:, 0] = value                      (test_class.py:249)
      mode="shift": z_which += value * (mu_lookup[1] - mu_lookup[0])   (notebook cells 11, 15)
    Returns (tokens (n, len(values), steps) int32, z0 (n,)) - all n*len(values) sequences decode as ONE batch.
    """
    was_training = model.training
    model.eval()
    try:
        dis_r, dis_n = model.encode(x)
        n, Z = dis_r.mean.shape
        if eps is None:
            eps = (torch.randn(n, Z), torch.randn(n, Z))      # repar() of test_class.py:53-56 draws r first, then n
        z_r = dis_r.mean + dis_r.stddev * eps[0].to(dis_r.mean.device)
        z_n = dis_n.mean + dis_n.stddev * eps[1].to(dis_r.mean.device)
        z0 = (z_r if which == "r" else z_n)[:, 0].clone()
        V = len(values)
        vals = torch.as_tensor(values, dtype=torch.float32, device=z_r.device)
        zr = z_r.unsqueeze(1).repeat(1, V, 1)
        zn = z_n.unsqueeze(1).repeat(1, V, 1)
        tgt = zr if which == "r" else zn
        if mode == "set":
            tgt[:, :, 0] = vals
        else:
            lk = model.mu_r_lookup if which == "r" else model.mu_n_lookup
            shift = lk.weight.data[1] - lk.weight.data[0]
            tgt += vals.view(1, V, 1) * shift.view(1, 1, Z)
        c = chroma.float().to(z_r.device).unsqueeze(1).repeat(1, V, 1)
        z = torch.cat([zr, zn, c], dim=2).view(n * V, -1)
        _, tok = greedy_decode(model, z, steps, want_logp=False)
        return tok.view(n, V, steps), z0
    finally:
        model.train(was_training)
