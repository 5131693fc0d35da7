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

    Bi <= Engine.single_launch_rows: ONE launch for the whole decode (fn_decode_greedy; above 32 rows a pipeline of 32- / 64-row blocks through
    its role workgroups).  Larger batches: steps x {layer-1 cell, W_ih2 projection,
    layer-2 cell, output GEMM, log_softmax+argmax} (from Engine.cell_decode_rows sequences on: steps x {layer-1 cell, layer-2 cell incl.
    its projection - fn_gru_cell_f32 -, output GEMM, argmax}; with want_logp=False: {layer-1 cell, layer-2 cell, output layer with the
    argmax in its epilogue - fn_out_argmax_f32}) captured once per (Bi, steps) into a hipGraph and replayed.  The captured
    kernels read the parameters and the engine's weight images IN PLACE (stable addresses, refreshed by Engine.refresh_weights
    after every optimiser step / load_state_dict), so a graph stays valid when the weights change."""
    eng = model.engine()
    z = z.float().contiguous()
    if _single_launch_ok(eng, z):
        res = _decode_single_launch(eng, z, steps, want_logp)
        if res is not None:
            return res
    if use_graph is None:
        use_graph = z.is_cuda
    if not use_graph:
        return _decode_body(eng, z, steps, want_logp, None, None)
    cache = eng.__dict__.setdefault("_decode_graphs", {})
    key = (z.shape[0], steps, bool(want_logp), z.shape[0] >= eng.cell_decode_rows, bool(getattr(eng, "fused_argmax", True)),
           bool(getattr(eng.ops, "dw_x6", False) and getattr(eng.ops, "cell_x6", False)))     # the captured launches depend on the path taken (and on the cells' arithmetic)
    ent = cache.get(key)
    if ent is None:
        zs = z.clone()
        tokens = torch.zeros(z.shape[0], steps, dtype=torch.int32, device=z.device)
        logp = torch.empty(z.shape[0], steps, E_VOCAB, device=z.device) if want_logp else None
        _decode_body(eng, zs, min(steps, 2), want_logp, logp, tokens, alloc_steps=steps)      # warm-up: allocates every buffer at its FINAL size (nothing is allocated inside the capture)
        g = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        getattr(eng.ops, "begin_capture", lambda: None)()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            _decode_body(eng, zs, steps, want_logp, logp, tokens)
        ent = cache[key] = (g, zs, logp, tokens)
    g, zs, logp, tokens = ent
    zs.copy_(z)
    g.replay()
    return (None if logp is None else logp.clone()), tokens.clone()


def _single_launch_ok(eng, z):
    """small batches decode as ONE launch (fn_decode_greedy: weight slices resident in LDS, activations handed over through L2)"""
    lo, hi = getattr(eng, "single_launch_skip", (0, -1))
    return (hasattr(eng.ops, "decode_greedy") and z.is_cuda and z.shape[0] <= eng.single_launch_rows and not (lo <= z.shape[0] <= hi)
            and eng.H <= 512 and eng.single_launch_decode)


def _decode_single_launch(eng, z, steps, want_logp):
    """None when the library reports the configuration as not eligible (e.g. fewer CUs than role workgroups) or the launch
    timed out waiting for a hand-over: the caller then takes the per-token path."""
    ops, P, H = eng.ops, eng.p, eng.H
    Bi = z.shape[0]
    h0g = eng.buf("dec_h0g", (Bi, H))
    ops.gemm(z, P["linear_init_global.weight"], h0g, bias=P["linear_init_global.bias"])
    rbg = eng.buf("dec_rbg", (Bi, 3 * H))
    ops.gemm(z, P["grucell_g.weight_ih"][:, E_VOCAB:], rbg)
    tokens = torch.zeros(Bi, steps, dtype=torch.int32, device=z.device)
    logp = torch.empty(Bi, steps, E_VOCAB, device=z.device) if want_logp else None
    ok = ops.decode_greedy(Bi, steps, H, E_VOCAB, E_VOCAB - 1, eng.whh_f["g"], P["grucell_g.bias_hh"], P["grucell_g.bias_ih"], eng.tab["g"], rbg, h0g,
                           eng.packs["ih2"], P["grucell_g_2.bias_ih"], eng.whh_f["g2"], P["grucell_g_2.bias_hh"], eng.packs["out"],
                           P["linear_out_g.bias"], tokens, logp)
    if not ok:
        return None
    if ops.gru_sync_error(clear=True):           # bounded spin gave up (another kernel held the CUs): results are garbage
        import warnings
        warnings.warn("single-launch greedy decode timed out waiting for a hand-over; repeating on the per-token kernels")
        return None
    return logp, tokens


def _decode_body(eng, z, steps, want_logp, logp, tokens, alloc_steps=None):
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
    if Bi >= eng.cell_decode_rows:
        # thousands of rows: every cell is ONE MFMA launch with the gates in its epilogue (fn_gru_cell_f32: LDS-free loop above 512 rows); layer 2 takes its input
        # projection in the same K loop - 3 launches + argmax per token instead of 4 + argmax, and no [B][3H] round trip
        # tokens only (the evaluators' sweeps): the output layer takes the argmax into its epilogue (fn_out_argmax_f32: packed (logit, column)
        # words by 64-bit atomic max, no logits, no argmax launch) and the next layer-1 cell reads its token from the packed word -
        # 3 launches per token; the int32 tokens are unpacked once at the end
        fused = not want_logp and getattr(eng, "fused_argmax", True) and hasattr(ops, "out_argmax")
        best = eng.buf("dec_best", (max(steps, alloc_steps or 0), Bi), dtype=torch.int64)[:steps] if fused else None
        if fused:
            best.zero_()
        for i in range(steps):
            cur, prv = i & 1, (i & 1) ^ 1
            tok_src = dict(idx_best=best[i - 1], best_v=E_VOCAB) if fused and i > 0 else dict(idx=tokens[:, i - 1] if i > 0 else None)
            ops.gru_cell(h0g if i == 0 else hx0[prv][0], P["grucell_g.weight_hh"], P["grucell_g.bias_hh"], hx0[cur][0], b_ih=P["grucell_g.bias_ih"],
                         gx_table=eng.tab["g"], start_token=E_VOCAB - 1, gx_rowbias=rbg, **tok_src)
            ops.gru_cell(hx0[cur][0] if i == 0 else hx1[prv][0], P["grucell_g_2.weight_hh"], P["grucell_g_2.bias_hh"], hx1[cur][0],
                         x=hx0[cur][0], w_ih=P["grucell_g_2.weight_ih"], b_ih=P["grucell_g_2.bias_ih"])
            if fused:
                ops.out_argmax(hx1[cur][0], P["linear_out_g.weight"], P["linear_out_g.bias"], best[i])
            else:
                ops.gemm(hx1[cur][0], P["linear_out_g.weight"], logits[:, :E_VOCAB], bias=P["linear_out_g.bias"])
                ops.vocab_argmax(logits, E_VOCAB, logp[:, i, :] if want_logp else None, tokens[:, i])
        if fused:
            ops.best_tokens(best, E_VOCAB, tokens)
        return logp, tokens
    for i in range(steps):
        cur, prv = i & 1, (i & 1) ^ 1
        ops.gru_seq_fwd([dict(B=Bi, T=1, H=H, w_hh_frag=eng.whh_f["g"], b_hh=P["grucell_g.bias_hh"], b_ih=P["grucell_g.bias_ih"],
                              h0=h0g if i == 0 else hx0[prv][0], h0_frag=None if i == 0 else hf0[prv], h_last_frag=hf0[cur],
                              gx_table=eng.tab["g"], idx=tokens, idx_shift=i - 1,
                              start_token=E_VOCAB - 1, gx_rowbias=rbg, h_all=hx0[cur])], persistent=False)
        ops.gemm(hx0[cur][0], P["grucell_g_2.weight_ih"], gx2[0], bias=P["grucell_g_2.bias_ih"])
        ops.gru_seq_fwd([dict(B=Bi, T=1, H=H, w_hh_frag=eng.whh_f["g2"], b_hh=P["grucell_g_2.bias_hh"],
                              h0=hx0[cur][0] if i == 0 else hx1[prv][0], h0_frag=hf0[cur] if i == 0 else hf1[prv], h_last_frag=hf1[cur],
                              gx_dense=gx2, h_all=hx1[cur])], persistent=False)
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
    lambda*shift-vector transfer (cells 11 + 15): every (sample, fader value) pair is one row of ONE decode batch.

    x (n, T) token ids or (n, T, 342) one-hot; chroma (n, 24); values: V fader values.
      mode="set":   z_which[:, 0] = value                                   (test_class.py:249, :298)
      mode="shift": z_which += value * (mu_lookup[1] - mu_lookup[0]);  which="both" moves z_r and z_n together, as the notebook does
    eps: (eps_r, eps_n), each (n, Z) - one draw per sample, shared by its V values - or (n, V, Z) - one draw per (sample, value),
    which is what V separate reference calls consume; None = drawn here (r first, then n).
    Returns (tokens (n, V, steps) int32, z0 (n,) or (n, V): the value of z_which[:, 0] before the change; which="both": z_r's)."""
    if which not in ("r", "n", "both") or mode not in ("set", "shift") or (which == "both" and mode == "set"):
        raise ValueError("which in {r, n, both}, mode in {set, shift}; 'both' only with mode='shift'")
    was_training = model.training
    model.eval()
    try:
        dis_r, dis_n = model.encode(x)
        n, Z = dis_r.mean.shape
        V = len(values)
        dev = dis_r.mean.device
        if eps is None:
            eps = (torch.randn(n, Z), torch.randn(n, Z))      # repar() of test_class.py:53-56 draws r first, then n
        er, en = (e.to(dev).float() for e in eps)
        per_value = er.dim() == 3
        if not per_value:
            er, en = er.unsqueeze(1).expand(n, V, Z), en.unsqueeze(1).expand(n, V, Z)
        zr = dis_r.mean.unsqueeze(1) + dis_r.stddev.unsqueeze(1) * er          # (n, V, Z), fresh tensors
        zn = dis_n.mean.unsqueeze(1) + dis_n.stddev.unsqueeze(1) * en
        z0 = (zn if which == "n" else zr)[:, :, 0].clone()
        if not per_value:
            z0 = z0[:, 0]
        vals = torch.as_tensor(values, dtype=torch.float32, device=dev)
        if mode == "set":
            (zr if which == "r" else zn)[:, :, 0] = vals
        else:
            for tgt, lk, on in ((zr, model.mu_r_lookup, which in ("r", "both")), (zn, model.mu_n_lookup, which in ("n", "both"))):
                if on:
                    tgt += vals.view(1, V, 1) * (lk.weight.data[1] - lk.weight.data[0]).view(1, 1, Z)
        c = chroma.float().to(dev).unsqueeze(1).expand(n, V, chroma.shape[-1])
        z = torch.cat([zr, zn, c], dim=2).reshape(n * V, -1)
        _, tok = greedy_decode(model, z, steps, want_logp=False)
        return tok.view(n, V, steps), z0
    finally:
        model.train(was_training)
