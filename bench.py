#!/usr/bin/env python3
"""Headline benchmark: event-tokens/sec of one full MusicAttrRegGMVAE training step (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--mode train|decode]

--gpus N > 1 without a launcher environment re-executes itself under ``torch.distributed.run`` (one rank per GPU, RCCL); when the
driver launches it under ``torch.distributed.run`` already, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* are taken from the environment.

mode train (default; BASELINE configs[1], configs[2-3] with N>1): a "step" = forward + every loss term + backward + gradient
  all-reduce (N>1) + clip + Adam on one synthetic minibatch (trainer_gmm.py:220-258 semantics), inputs resident in HBM.  Workload:
  hidden 512, z 128, K=2, B=256 sequences per GPU (weak scaling), T=256 event tokens, Tr=64 rhythm/note steps, fp32.
mode decode (BASELINE configs[4]): a "step" = encode 256 sequences (T=256), 8 fader values each on z_r[:, 0], greedy decode of the
  2048 rows for 300 steps (test_class.py:233-254 batched; replicas only, no collective).

--arith f32 | bf16x6: the arithmetic of the deep MFMA products (music-fader-nets_amd/arith.py; default = the package default).  Both are fp32-class
(24-bit operands, fp32 accumulation); the line says which one `value` was measured on (`arith`) and carries the other one beside it
(`fp32_mfma_value` / `bf16x6_value`, timed in a child process on the same seeds).  A bf16 x 6 kernel is rated against the dense bf16 MFMA peak / 6
(416.7 fp32-equivalent TFLOP/s), an fp32-MFMA kernel against 157.3.

Prints ONE JSON line on rank 0 with the contract fields plus `roofline` (the kernel SYMBOL - template instances merged, as rocprofv3 --stats
lists them - with the largest time per step), `roofline_by_symbol`, `roofline_scans` (the four scan rows as one time-weighted figure),
`roofline_all` (split by launch shape), `sustained_ms_per_step` (200 more steps behind the timed region), (N=1) `cpu_baseline`, `decode`
(the configs[4] measurement, so that the default run records it too) and, with collectives in place, `comm` (per-bucket all-reduce time
and how long the step's stream stood still for them).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H, Z, K, B, T, TR = 512, 128, 2, 256, 256, 64
DEC_SEQS, DEC_VALUES, DEC_STEPS = 256, 8, 300
# algorithmic work (SURVEY.md 8d / DESIGN.md)
FLOP_PER_SAMPLE_STEP = 2.0 * H * 3 * H                 # recurrent part of one GRU step of one scan for one sample: 1.573 MFLOP
F_ALG_PER_TOKEN = 37.12e6                              # whole training step, per event token
F_ALG_DECODE_PER_TOKEN = 5.07e6                        # one greedy decode step of one sequence (token projection = row gather)
DECODE_WEIGHT_BYTES = 10.14e6                          # weights one decode step touches (W_hh_g, W_ih_g2, W_hh_g2, W_out)
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16X6_TFLOPS = 2500.0 / 6.0                      # fp32-equivalent peak of a bf16 x 6 kernel: dense bf16 MFMA peak (2.5 PFLOP/s) / 6 partial products = 416.7
PEAK_HBM_GBS = 8000.0
PEAK_OF = {"f32": PEAK_F32_MFMA_TFLOPS, "bf16x6": PEAK_BF16X6_TFLOPS}


class TimedOps:
    """Proxy around HipOps that brackets every op call with HIP events ON THE STREAM THE OP IS LAUNCHED ON (the engine enters the
    lane's stream before calling the op, so torch's current stream is that stream).  Used for the per-kernel roofline numbers only,
    in eager passes outside the timed region."""

    def __init__(self, ops):
        object.__setattr__(self, "_ops", ops)
        object.__setattr__(self, "records", [])

    def __getattr__(self, name):
        attr = getattr(self._ops, name)
        if not callable(attr) or name in ("stream", "workspace", "gates_floats", "frag_floats", "gru_sync_error", "_frag_ws", "_sync_ws"):
            return attr

        def call(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            res = attr(*a, **k)
            e1.record()
            self.records.append((name, a, k, e0, e1, _x6_of(self._ops, name, a, k)))
            return res
        return call

    def __setattr__(self, name, value):
        setattr(self._ops, name, value)


def _x6_of(ops, name, a, k):
    """which arithmetic this op call ran on (mirrors hipops.py's choices)"""
    if not getattr(ops, "dw_x6", False):
        return False
    if name == "gru_seq_fwd":
        return bool(k.get("x6", None) if k.get("x6", None) is not None else ops.gru_fwd_x6_ok(a[0]))
    if name == "gru_seq_bwd":
        return bool(k.get("x6", None) if k.get("x6", None) is not None else ops.gru_bwd_x6_ok(a[0]))
    # weight-gradient products: which of the bf16 x 6 kernels (hipops._x6_mode: 128 x 256 tiles for the T*B-deep products, 128 x 128 otherwise)
    from music_fader_nets_amd import _lib
    if name == "gru_dwhh" and a[2].shape[0] >= 1024:
        fl = ops._x6_mode(k.get("splitk", 1), a[2].shape[1], a[2].shape[0], _lib.GEMM_BF16X6)[1]
    elif name == "gemm" and (not k.get("a_k", True)) and (not k.get("b_k", True)) and a[0].shape[0] >= 1024:
        fl = ops._x6_mode(k.get("splitk", 1), a[2].shape[1], a[0].shape[0], _lib.GEMM_BF16X6)[1]
    elif name == "gemm" and k.get("a_k", True) and k.get("b_k", True):
        M, N = a[2].shape
        Kk = a[0].shape[1]
        ok = (getattr(ops, "nt_x6", False) and k.get("splitk", 1) <= 1 and M % 128 == 0 and N % 128 == 0 and Kk % 32 == 0 and Kk >= 128 and (M // 128) * (N // 128) >= 128)
        return "gemm_nt_x6w_kernel" if ok else False
    else:
        return False
    return "gemm_tn_x6_kernel" if fl & _lib.GEMM_X6_PERWAVE else ("gemm_tn_x6v_kernel" if fl & _lib.GEMM_X6_WIDE else "gemm_tn_x6w_kernel")


def _classify(name, a, k):
    """op call -> (roofline row, work of that launch) or None"""
    if name in ("gru_seq_fwd", "gru_seq_bwd"):
        scans = a[0]
        work = sum(s["B"] * s["T"] for s in scans) * FLOP_PER_SAMPLE_STEP
        kind = "fwd" if name == "gru_seq_fwd" else "bwd"
        if len(scans) == 4:
            return "enc_%s_scan" % kind, work
        if all(str(s_.get("tag", "")).startswith(("dec_l", "sd_")) for s_ in scans):
            return ("dec_%s_scan_chunk" % kind, work) if len(scans) == 2 else None      # steady state: layer 1 + layer 2 in one launch
        if len(scans) == 2:
            return "subdec_%s_scan" % kind, work
        return None
    if name == "gru_dwhh":
        rows, Hh = a[2].shape
        # two kernel symbols: the <= 128-register instance (decoder side, beside the encoder backward scan) and the full one
        if rows < 4096:
            return None
        row = "dwhh_gemm_tn_lean" if k.get("lean") else ("dwhh_gemm_tn" if rows >= 32768 else "dwhh_gemm_tn_attr")
        return row, 2.0 * rows * 3 * Hh * Hh
    if name == "embed_grad_sorted":                        # HBM-bound: every gate-gradient row of every job is read once
        nbytes = float(sum(j["dgx"].numel() for j in a[1]) * 4)
        return ("embed_grad", nbytes) if nbytes >= 1e8 else None
    if name == "gemm":
        A, Bm, Cm = a[0], a[1], a[2]
        M, N = Cm.shape
        Kk = A.shape[1] if k.get("a_k", True) else A.shape[0]
        if 2.0 * M * N * Kk >= 2e10:
            return ("gemm_nt" if k.get("b_k", True) else ("gemm_nn" if k.get("a_k", True) else "gemm_tn")), 2.0 * M * N * Kk
        return None
    if name == "out_head":                                 # fused output head: the projection's flops (the softmax rides in its epilogue)
        h, W = a[0], a[1]
        return "out_head", 2.0 * h.shape[0] * W.shape[0] * h.shape[1]
    if name == "vocab_logsoftmax":
        return "out_head_softmax", None
    return None


def _symbol(name, a, k, x6=False):
    """op call -> (kernel symbol as rocprofv3 prints it without template arguments, bound, work of that launch) for EVERY launch of the heavy
    symbols, whatever its shape (roofline_by_symbol merges what `_classify` splits by launch shape)"""
    if name in ("gru_seq_fwd", "gru_seq_bwd"):
        work = sum(s["B"] * s["T"] for s in a[0]) * FLOP_PER_SAMPLE_STEP
        return (("gru_fwd_x6pp_kernel" if x6 else "gru_fwd_pp_kernel") if name == "gru_seq_fwd" else ("gru_bwd_x6_kernel" if x6 else "gru_bwd_rs_kernel")), "mfma", work
    if name == "gru_dwhh":
        rows, Hh = a[2].shape
        return (x6 if x6 else "gemm_tn_kernel"), "mfma", 2.0 * rows * 3 * Hh * Hh
    if name == "gemm":
        A, Cm = a[0], a[2]
        M, N = Cm.shape
        Kk = A.shape[1] if k.get("a_k", True) else A.shape[0]
        if k.get("a_k", True) and k.get("b_k", True):
            sym = x6 if x6 else "gemm_nt_direct_kernel / gemm_kernel"
        else:
            sym = "gemm_kernel" if k.get("a_k", True) else (x6 if x6 else "gemm_tn_kernel")
        return sym, "mfma", 2.0 * M * N * Kk
    if name == "out_head":
        h, W = a[0], a[1]
        return "out_head_kernel", "mfma", 2.0 * h.shape[0] * W.shape[0] * h.shape[1]
    if name == "embed_grad_sorted":
        return "eg_piece_kernel + eg_final_kernel", "hbm", float(sum(j["dgx"].numel() for j in a[1]) * 4)
    return None


SYMBOL_NOTE = {
    "gru_fwd_x6pp_kernel": "forward weight-stationary scans on the bf16 MFMA (exact bf16 triple splits, 6 products), ping-pong over two row halves (all launches: encoder 4 x 256 rows x 256 steps, decoder pipeline chunks, attribute decoders); rated against 2.5 PFLOP/s / 6",
    "gru_bwd_x6_kernel": "backward weight-stationary scans on the bf16 MFMA (gate gradients exchanged as exact bf16 triples, W_hh^T slice in AGPRs + LDS; the ENCODER launch only - the 32-row groups of the decoder pipeline / attribute decoders stay on gru_bwd_rs_kernel<1>); rated against 2.5 PFLOP/s / 6",
    "gemm_tn_x6v_kernel": "weight-gradient products dW = dY^T X on the bf16 MFMA, 128 x 256 output tiles: producer wavefronts split every operand value once per workgroup into LDS, consumer wavefronts only multiply (the T*B-deep products: dW_hh of the encoder directions / decoder layers via fn_gru_dwhh_f32, W_ih2, output layer); rated against 2.5 PFLOP/s / 6",
    "gemm_tn_x6w_kernel": "the same on 128 x 128 output tiles (the products over Tr*B rows: attribute decoders); rated against 2.5 PFLOP/s / 6",
    "gemm_nt_x6w_kernel": "Linear-forward / dX products of the decoder pipeline on the bf16 MFMA (gx2 = hx0 W_ih2^T and dhx0 = dgx2 W_ih2 per 32-step chunk, dhx1 = dlogits W_out), producer / consumer form; rated against 2.5 PFLOP/s / 6",
    "gemm_tn_x6_kernel": "the round-5 kernel (every wavefront splits its own operands; only with HipOps.x6_perwave)",
    "gru_fwd_pp_kernel": "forward weight-stationary scans, ping-pong over two row halves (all launches: encoder 4 x 256 rows x 256 steps, decoder pipeline chunks, attribute decoders)",
    "gru_bwd_rs_kernel": "backward weight-stationary scans on the fp32 MFMA, W_hh^T slice half register-stationary (arithmetic f32: all launches; bf16x6: the decoder pipeline chunks and attribute decoders)",
    "gemm_tn_kernel": "weight-gradient products dW = dY^T X (dW_hh of every scan via fn_gru_dwhh_f32, dW of the dense layers)",
}


X6_KERNEL = {  # rows whose launches run on the bf16 x 6 kernels when that arithmetic is chosen
    "enc_fwd_scan": "gru_fwd_x6pp_kernel<1> (4 encoder scans x 256 steps, one launch; bf16 MFMA, exact triple splits)",
    "dec_fwd_scan_chunk": "gru_fwd_x6pp_kernel<2> (one launch of the decoder pipeline: 2 scans x 256 rows x 32 steps; bf16 MFMA, exact triple splits)",
    "subdec_fwd_scan": "gru_fwd_x6pp_kernel (both sub-decoders, 64 steps)",
    "enc_bwd_scan": "gru_bwd_x6_kernel<2> (4 encoder scans x 256 steps, one launch; bf16 MFMA, exact triple splits)",
    "dwhh_gemm_tn": "gemm_tn_x6v_kernel via fn_gru_dwhh_f32 (dW_hh of an encoder direction / a decoder layer: [3H x T*B] x [T*B x H], 24 tiles of 128 x 256 x 32 K ranges; 6 launches per step)",
    "dwhh_gemm_tn_attr": "gemm_tn_x6w_kernel via fn_gru_dwhh_f32 (dW_hh of the attribute decoders, K = Tr*B rows, 48 tiles of 128 x 128 x 16 K ranges)",
    "gemm_tn": "gemm_tn_x6v_kernel (dW of dense layers: W_ih2, output layer)",
    "gemm_nt": "gemm_nt_x6w_kernel (dhx1 = dlogits W_out through the transposed weight image)",
}
ROW_INFO = {   # row -> (bound, unit of work, kernel, peak share of the chip)
    "enc_fwd_scan": ("mfma", "flop", "gru_fwd_pp_kernel<1> (4 encoder scans x 256 steps, one launch)", 1.0),
    "enc_bwd_scan": ("mfma", "flop", "gru_bwd_rs_kernel<2> (4 encoder scans x 256 steps, one launch)", 1.0),
    "dec_fwd_scan_chunk": ("mfma", "flop", "gru_fwd_pp_kernel<2> (one launch of the decoder pipeline: 2 scans x 256 rows x 32 steps - layer 1 chunk k + layer 2 chunk k-2, attribute-decoder chunks at both ends)", 1.0),
    "dec_bwd_scan_chunk": ("mfma", "flop", "gru_bwd_rs_kernel<1> (one launch of the decoder pipeline: 2 scans x 256 rows x 32 steps - layer 2 chunk k + layer 1 chunk k+2, attribute-decoder chunks at both ends)", 1.0),
    "subdec_fwd_scan": ("mfma", "flop", "gru_fwd_pp_kernel (both sub-decoders, 64 steps)", 1.0),
    "subdec_bwd_scan": ("mfma", "flop", "gru_bwd_rs_kernel (both sub-decoders, 64 steps)", 1.0),
    "dwhh_gemm_tn": ("mfma", "flop", "gemm_tn_kernel via fn_gru_dwhh_f32 (dW_hh of an encoder direction / a decoder layer: [3H x T*B] x [T*B x H], 48 tiles x 16 K ranges; 6 launches per step)", 1.0),
    "dwhh_gemm_tn_attr": ("mfma", "flop", "gemm_tn_kernel via fn_gru_dwhh_f32 (dW_hh of the attribute decoders, K = Tr*B rows, 48 tiles x 16 K ranges)", 1.0),
    "dwhh_gemm_tn_lean": ("mfma", "flop", "gemm_tn_lean_kernel via fn_gru_dwhh_f32 (dW_hh of the decoder-side scans, <= 128 registers)", 1.0),
    "gemm_tn": ("mfma", "flop", "gemm_tn_kernel (dW of dense layers)", 1.0),
    "gemm_nt": ("mfma", "flop", "gemm_kernel (X W^T: W_ih2 projection, output layer)", 1.0),
    "gemm_nn": ("mfma", "flop", "gemm_kernel (dY W: input gradients)", 1.0),
    "out_head": ("mfma", "flop", "out_head_kernel (512 -> 342 projection + log-softmax + NLL + gradient seed, logits never written)", 1.0),
    "embed_grad": ("hbm", "bytes", "eg_piece_kernel + eg_final_kernel: token-segment sums of the gate-gradient rows (embed.hip)", 1.0),
}


# HBM bytes per launch come from rocprofv3 --pmc passes of the same launches (2 x FETCH_SIZE with the gfx950 correction + WRITE_SIZE, separate passes;
# scratch/r5_pmc_step.sh writes the per-row / per-symbol means into profiles/r05_pmc_traffic.json).  bench.py cannot run the profiler around itself,
# so these are STATIC, committed measurements: the line says so (`traffic_static`) and names the file.
PMC_FILE = next((f for f in (os.path.join(ROOT, "profiles", n) for n in ("r06_pmc_traffic.json", "r05_pmc_traffic.json")) if os.path.exists(f)),
                os.path.join(ROOT, "profiles", "r05_pmc_traffic.json"))


def pmc_traffic():
    try:
        d = json.load(open(PMC_FILE))
        return d.get("by_row", {}), d.get("by_symbol", {}), "profiles/%s (%s)" % (os.path.basename(PMC_FILE), d.get("how", "rocprofv3 --pmc"))
    except (OSError, ValueError):
        return {}, {}, None


def roofline_rows(records):
    """rows of `roofline_all`: one per (launch shape, arithmetic) - a row whose launches ran on both arithmetics is split into `row` (the arithmetic
    of most of its time) and `row[other]`, never mislabelled"""
    agg = {}
    for name, a, k, e0, e1, x6 in records:
        c = _classify(name, a, k)
        if c is None or c[1] is None or c[0] not in ROW_INFO:
            continue
        ms = e0.elapsed_time(e1)
        ent = agg.setdefault((c[0], bool(x6)), [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += ms
        ent[2] += c[1]
    rows = {}
    for (row, x6), (cnt, ms, work) in agg.items():
        bound, unit, kernel, share = ROW_INFO[row]
        arith = "bf16x6" if x6 else "f32"
        other = agg.get((row, not x6))
        key = row if other is None or other[1] < ms else "%s[%s]" % (row, arith)
        if bound == "mfma":
            ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_OF[arith] * share, "TFLOP/s"
            if x6:
                kernel = X6_KERNEL.get(row, kernel + " [launches that ran on the bf16 x 6 kernels]")
        else:
            ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS * share, "GB/s"
        rows[key] = dict(bound=bound, kernel=kernel, launches=cnt, avg_launch_us=round(ms / cnt * 1e3, 1), achieved=round(ach, 2),
                         peak=round(peak, 1), unit=u, frac=round(ach / peak, 4), work_per_launch=work / cnt, total_us=ms * 1e3)
        if bound == "mfma":
            rows[key]["arith"] = arith
            rows[key]["frac_of_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
    return rows


def symbol_rows(records, reps):
    agg = {}
    for name, a, k, e0, e1, x6 in records:
        c = _symbol(name, a, k, x6)
        if c is None:
            continue
        ent = agg.setdefault(c[0], [c[1], 0, 0.0, 0.0])
        ent[1] += 1
        ent[2] += e0.elapsed_time(e1)
        ent[3] += c[2]
    rows = {}
    for sym, (bound, cnt, ms, work) in agg.items():
        x6 = "_x6" in sym
        if bound == "mfma":
            ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_OF["bf16x6" if x6 else "f32"], "TFLOP/s"
        else:
            ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        rows[sym] = dict(bound=bound, kernel=sym, note=SYMBOL_NOTE.get(sym), launches_per_step=round(cnt / reps, 1), avg_launch_us=round(ms / cnt * 1e3, 1),
                         achieved=round(ach, 2), peak=round(peak, 1), unit=u, frac=round(ach / peak, 4), work_per_launch=work / cnt,
                         us_per_step=round(ms * 1e3 / reps, 1))
        if bound == "mfma":
            rows[sym]["arith"] = "bf16x6" if x6 else "f32"
            rows[sym]["frac_of_fp32_mfma_peak"] = round(ach / PEAK_F32_MFMA_TFLOPS, 4)
    return rows


SCAN_ROWS = ("enc_fwd_scan", "enc_bwd_scan", "dec_fwd_scan_chunk", "dec_bwd_scan_chunk")


def scans_row(rows):
    """the four scan rows of roofline_all as ONE time-weighted figure (0.633 then)"""
    have = [rows[r] for r in SCAN_ROWS if r in rows]
    if not have:
        return None
    us = sum(r["us_per_step"] for r in have)
    work = sum(r["work_per_launch"] * r["launches"] for r in have) / max(1, have[0]["_reps"])
    ach = work / (us * 1e-6) / 1e12
    # time-weighted over rows of possibly different arithmetic: every row's time is set against ITS peak (sum of row time x row fraction / time)
    frac = sum(r["us_per_step"] * r["frac"] for r in have) / us
    return dict(bound="mfma", rows=[r for r in SCAN_ROWS if r in rows], us_per_step=round(us, 1), achieved=round(ach, 2), unit="TFLOP/s",
                frac=round(frac, 4), frac_of_fp32_mfma_peak=round(ach / PEAK_F32_MFMA_TFLOPS, 4), arith={r: rows[r].get("arith") for r in SCAN_ROWS if r in rows},
                flop_per_step=work)


def per_kernel_rooflines(trainer, batch, eps, reps=5):
    """`reps` eager fwd+bwd passes (no optimiser update) under the TimedOps proxy, all lanes serialised on one stream -> per-kernel
    average launch durations with each kernel running alone (mean of all launches, not the best) and their roofline fractions."""
    eng = trainer.model.engine()
    real = eng.ops
    proxy = TimedOps(real)
    eng.ops = proxy
    eng.serialize_lanes = True          # every lane on one stream: each kernel is timed running ALONE (as under rocprofv3 --pmc);
    try:                                # in the real step GEMMs / segment sums overlap each other and the scans' launch tails
        for _ in range(reps):
            trainer.loss_and_grads(20000, batch, eps)
        torch.cuda.synchronize()
    finally:
        eng.ops = real
        eng.serialize_lanes = False
    if real.gru_sync_error():
        raise RuntimeError("weight-stationary scan: a workgroup gave up waiting (sync error flag set)")
    rows = roofline_rows(proxy.records)
    for r in rows.values():
        r["us_per_step"] = round(r.pop("total_us") / reps, 1)        # kernel time of this row in ONE training step
        r["_reps"] = reps
    return rows, symbol_rows(proxy.records, reps)


def _max_over_ranks(x, dev):
    """MAX over ranks of a host scalar through the control-plane process group (gloo: CPU tensor; nccl: device tensor)"""
    on_dev = torch.distributed.get_backend() == "nccl"
    t = torch.tensor([x], dtype=torch.float64, device=dev if on_dev else "cpu")
    torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
    return float(t.item())


def comm_times(trainer, ctx, batch, eps, step, reps=3):
    """per-bucket all-reduce durations and the time the step's stream waits for the last bucket, from `reps` EAGER steps (events on
    the communicator's stream; a captured step cannot carry timing events)"""
    use_graph = trainer.use_graph
    trainer.use_graph = False
    ctx.timing = {}
    try:
        for i in range(reps):
            trainer.step_device(step + i, batch, eps)
        torch.cuda.synchronize()
        out = {}
        for key, name in (("bucket1", "bucket1_ms"), ("bucket2", "bucket2_ms"), ("bucket3", "bucket3_ms"), ("exposed", "exposed_ms")):
            evs = ctx.timing.get(key, [])
            out[name] = round(float(np.mean([a.elapsed_time(b) for a, b in evs])), 4) if evs else None
        out["bucket1_MB"] = round(trainer.flat.bucket_split * 4 / 1e6, 2)
        out["bucket2_MB"] = round((trainer.flat.bucket_split2 - trainer.flat.bucket_split) * 4 / 1e6, 2)
        out["bucket3_MB"] = round((trainer.flat.n - trainer.flat.bucket_split2) * 4 / 1e6, 2)
        out["note"] = ("bucket 1 (decoder-side gradients) is issued behind the decoder weight-gradient GEMMs of the side lane (they execute once the encoder "
                       "backward scan has ended) and runs beside the encoder-side weight-gradient GEMMs; bucket 2 (heads, component means, rhythm encoder) beside the note encoder's weight-gradient GEMMs; bucket 3 (note "
                       "encoder) is the exposed one; exposed = wait of the step's stream in front of clip+Adam")
        return out
    finally:
        ctx.timing = None
        trainer.use_graph = use_graph


def respawn_under_launcher(n):
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-run this command as N ranks on this node."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


def make_inputs(trainer, rank, world):
    """the GLOBAL batch of the job is ONE draw (RandomState(0), B * world rows) and every rank keeps its rows; the noise is drawn for the global batch
    from one seed on every rank and sliced (GMVAETrainer.draw_eps): an N-rank step IS the single-process step of the global batch (rank 0 of a
    1-GPU run: the batch of tests/golden/c1.npz)"""
    from music_fader_nets_amd.synth import synth_batch
    b = synth_batch(np.random.RandomState(0), B * world, T, TR)
    lo, hi = rank * B, (rank + 1) * B
    batch = trainer.prepare_batch(b["d"][lo:hi], b["r"][lo:hi], b["n"][lo:hi], b["c"][lo:hi], b["r_density"][lo:hi], b["n_density"][lo:hi])
    torch.manual_seed(99)
    eps = trainer.draw_eps(B, T)                            # the reference's draw order; the same eps every step
    return batch, eps


def dp_selfcheck(pkg, ctx, dev, rank, world, log):
    """N ranks through RCCL == one process on the global batch, at a small size (hidden 64, 8 rows per rank, T = 16): every rank takes one
    data-parallel step, rank 0 repeats it alone on the concatenated batch; the first-step loss tuples must agree (trainer_gmm.py:249-251: the
    reduced gradient is what clip + Adam see - compared through the loss of a SECOND step, which depends on the first update)"""
    from music_fader_nets_amd.synth import synth_batch
    Hs, Zs, Bs, Ts, Trs = 64, 32, 8, 16, 8
    b = synth_batch(np.random.RandomState(7), Bs * world, Ts, Trs)

    def run(c, lo, hi, gen_rows):
        torch.manual_seed(4321)
        m = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, Hs, Zs, 32, n_component=K).to(dev)
        tr = pkg.GMVAETrainer(m, lr=1e-3, beta=0.2, dist_ctx=c)
        tr.use_graph = False
        batch = tr.prepare_batch(b["d"][lo:hi], b["r"][lo:hi], b["n"][lo:hi], b["c"][lo:hi], b["r_density"][lo:hi], b["n_density"][lo:hi])
        out = []
        for it in range(2):
            torch.manual_seed(55 + it)
            eps = tr.draw_eps(hi - lo, Ts)
            beta0, Bg = tr.step_device(20000 + it, batch, eps)
            out.append(tr._tuple8(beta0, Bg, False))
        return out
    dp = run(ctx, rank * Bs, (rank + 1) * Bs, Bs * world)
    res = None
    if rank == 0:
        single = run(None, 0, Bs * world, Bs * world)
        rel = max(abs(a - b_) / max(1e-12, abs(b_)) for ta, tb in zip(dp, single) for a, b_ in zip(ta, tb))
        res = dict(loss_dp=[round(t[0], 6) for t in dp], loss_single_process=[round(t[0], 6) for t in single], max_rel_diff_of_the_tuples=float("%.3g" % rel),
                   shape="hidden 64, %d x %d rows, T = %d, two steps" % (world, Bs, Ts))
        log("dp self-check: %d RCCL ranks vs one process on the global batch: max rel diff %.3g" % (world, rel))
        assert rel <= 2e-4, "data-parallel step deviates from the single-process step of the global batch: %r" % (res,)
    return res


def bench_train(args, pkg, ctx, local, rank, world, log):
    from music_fader_nets_amd import arith as arith_mod
    dev = torch.device("cuda", local)
    arith = arith_mod.resolve(args.arith)
    torch.manual_seed(1234)                                 # identical weights on every rank
    model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, H, Z, 32, n_component=K).to(dev)
    model.set_arith(arith)
    trainer = pkg.GMVAETrainer(model, lr=1e-3, beta=0.2, dist_ctx=ctx)
    batch, eps = make_inputs(trainer, rank, world)
    log("model + batch ready on %s, arithmetic %s" % (dev, arith))
    step = 20000
    first = None
    for i in range(args.warmup):
        beta0, Bg = trainer.step_device(step, batch, eps)
        if i == 0:
            first = trainer._tuple8(beta0, Bg, False)       # the very first optimisation step from the seeded init (one sync)
        step += 1
    torch.cuda.synchronize()
    if ctx is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step_device(step, batch, eps)
        step += 1
    t_host = time.perf_counter() - t0                       # host enqueue time (the GPU runs behind asynchronously)
    torch.cuda.synchronize()
    if ctx is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ctx is not None:
        dt = _max_over_ranks(dt, dev)
    tup = trainer._tuple8(0.2, B * world, False)            # one sync: the loss numbers of the last step (finite check)
    assert all(np.isfinite(tup)), tup
    tokens_per_s = world * B * T * args.steps / dt
    log("timed region done: %.3f ms/step (host enqueue %.3f ms/step)" % (dt / args.steps * 1e3, t_host / args.steps * 1e3))
    if args.leg_only:                                       # child process of the default run: the other arithmetic's timed region, nothing else
        return dict(arith=arith, ms_per_step=round(dt / args.steps * 1e3, 3), value=round(tokens_per_s, 1), unit="event-tokens/s", steps=args.steps,
                    first_step_loss=None if first is None else round(first[0], 4), last_loss=round(tup[0], 4))
    # sustained rate: >= 200 more replays right behind the timed region (clock / thermal settle: r03 soak 23.64 ms vs 22.78 in the 20-step bench)
    sustained = None
    if args.sustain > 0:
        torch.cuda.synchronize()
        if ctx is not None:
            torch.distributed.barrier()
        t1 = time.perf_counter()
        for _ in range(args.sustain):
            trainer.step_device(step, batch, eps)
            step += 1
        torch.cuda.synchronize()
        if ctx is not None:
            torch.distributed.barrier()
        sustained = (time.perf_counter() - t1) / args.sustain
        if ctx is not None:
            sustained = _max_over_ranks(sustained, dev)
        log("sustained: %.3f ms/step over %d more steps" % (sustained * 1e3, args.sustain))
    rows, by_symbol = per_kernel_rooflines(trainer, batch, eps)       # every rank: the pass contains the regulariser's all-gather
    log("per-kernel timing done")
    out = {
        "metric": "event-tokens/sec GM-VAE train, seq256 b256", "value": round(tokens_per_s, 1), "unit": "event-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # the arithmetic the path computes in: fp32 operands and fp32 accumulation in both cases; "f32/bf16x6" = every fp32 product formed exactly from bf16 pieces
        "dtype": "f32" if arith_mod.resolve(arith) == arith_mod.F32 else "f32/bf16x6", "arith": arith_mod.describe(arith), "data": "synthetic",
        "config": {"workload": "MusicAttrRegGMVAE train step (fwd+losses+bwd+clip+Adam), hidden 512, z 128, K=2, "
                               "B=256/GPU, T=256, Tr=64 (BASELINE configs[1]; N>1: DP, RCCL grad all-reduce)",
                   "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world},
        "last_loss": round(tup[0], 4),
    }
    if sustained is not None:
        out["sustained_ms_per_step"] = round(sustained * 1e3, 3)
        out["sustained_value"] = round(world * B * T / sustained, 1)
        out["sustained_steps"] = args.sustain
    if ctx is not None and getattr(ctx, "rccl", None) is not None:
        comm = comm_times(trainer, ctx, batch, eps, step)
        comm["rccl_ranks"], comm["rccl_rank"] = ctx.rccl.ranks()      # what RCCL itself says (ncclCommCount / ncclCommUserRank)
        assert comm["rccl_ranks"] == world, "RCCL reports %r ranks, the launcher %d" % (comm["rccl_ranks"], world)
        comm["dp_selfcheck"] = _aux(lambda: dp_selfcheck(pkg, ctx, dev, rank, world, log), log, "dp self-check") if world > 1 else None
        if rank == 0:
            out["comm"] = comm
    if rank == 0:
        # `roofline` = the kernel SYMBOL (template instances merged, as rocprofv3 --stats lists them) the step spends most of its time in;
        # `roofline_all` keeps the split by launch shape, `roofline_scans` the four scan rows as one time-weighted figure.  A bf16 x 6 kernel is
        # rated against the dense bf16 MFMA peak / 6 = 416.7 fp32-equivalent TFLOP/s, an fp32-MFMA kernel against 157.3.
        traffic_row, traffic_sym, traffic_src = pmc_traffic()
        scans = scans_row(rows)
        for r in rows.values():
            r.pop("_reps", None)
        dom_sym = max(by_symbol, key=lambda r: by_symbol[r]["us_per_step"])
        dom = dict(by_symbol[dom_sym])
        dom.update(symbol=dom_sym, traffic=traffic_sym.get(dom_sym), traffic_static=True, traffic_source=traffic_src if dom_sym in traffic_sym else None,
                   flop_per_launch=dom.pop("work_per_launch"),
                   step_frac=round(tokens_per_s / world * F_ALG_PER_TOKEN / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                   step_frac_note="whole step: 37.12 MFLOP per token x tokens/s against the fp32 MFMA peak (157.3 TFLOP/s), whatever arithmetic the kernels ran on")
        out["roofline"] = dom
        for r in by_symbol.values():
            r.pop("work_per_launch", None)
        out["roofline_by_symbol"] = by_symbol
        out["roofline_scans"] = scans
        for row, tr_ in traffic_row.items():
            if row in rows:
                rows[row]["traffic"], rows[row]["traffic_static"], rows[row]["traffic_source"] = tr_, True, traffic_src
        out["roofline_all"] = rows
        out["roofline_worst"] = min(rows, key=lambda r: rows[r]["frac"])
        if first is not None and world == 1:                # same seeds as tests/golden/c1.npz (the reference's own train() at this size)
            out["first_step_loss"] = round(first[0], 4)
            gpath = os.path.join(ROOT, "tests", "golden", "c1.npz")
            if os.path.exists(gpath):
                ref = float(np.load(gpath)["train_tuples"][0][0])
                out["first_step_loss_reference"] = round(ref, 4)
                assert abs(first[0] - ref) <= 5e-4 * abs(ref), "first optimisation step deviates from the reference: %r vs %r" % (first[0], ref)
    if world == 1 and not args.no_other_arith:
        # the OTHER arithmetic beside the headline: same seeds, same steps, its own process (whatever happens to that leg cannot take the
        # headline line down).  Headline on bf16 x 6 -> `fp32_mfma_value` / `fp32_mfma_ms_per_step`; headline on the fp32 MFMA -> `bf16x6_*`.
        other = "f32" if arith == "bf16x6" else "bf16x6"
        leg = _aux(lambda: other_arith_leg(args, other, log), log, "%s leg" % other)
        pfx = "fp32_mfma" if other == "f32" else "bf16x6"
        out[pfx + "_leg"] = leg
        if "error" not in leg:
            out[pfx + "_value"], out[pfx + "_ms_per_step"] = leg["value"], leg["ms_per_step"]
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        out["cpu_baseline"] = _aux(lambda: cpu_baseline.time_baseline(H, Z, B, T, TR), log, "cpu baseline")
    if world == 1 and not args.no_decode:
        # BASELINE configs[4] rides along in the default line (about 0.3 s of GPU time): 1 warm-up + 3 timed passes
        del trainer
        dargs = argparse.Namespace(steps=3, warmup=1, no_cpu_baseline=args.no_cpu_baseline, sustain=0)
        d = _aux(lambda: bench_decode(dargs, pkg, None, local, rank, world, log), log, "decode leg")
        out["decode"] = d if "error" in d else dict(metric=d["metric"], value=d["value"], unit=d["unit"], ms_per_pass=d["ms_per_step"],
                                                    workload=d["config"]["workload"], roofline=d["roofline"], cpu_baseline=d.get("cpu_baseline"))
    return out


def _aux(fn, log, what):
    """an auxiliary leg (reported beside the headline) must never take the headline line down with it"""
    try:
        return fn()
    except Exception as e:                        # noqa: BLE001 - whatever went wrong is reported in the line
        log("%s FAILED: %s: %s" % (what, type(e).__name__, e))
        return {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}


def other_arith_leg(args, arith, log):
    cmd = [sys.executable, os.path.abspath(__file__), "--arith", arith, "--leg-only", "--steps", str(args.steps), "--warmup", str(max(1, args.warmup)),
           "--sustain", "0", "--no-cpu-baseline", "--no-decode", "--no-other-arith"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FN_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    for line in r.stderr.decode(errors="replace").splitlines():
        if line.startswith("[bench") and "timed region" in line:
            log("(%s leg) %s" % (arith, line.split("] ", 1)[-1]))
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("child exited with %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:]))
    return json.loads(lines[-1])


def bench_decode(args, pkg, ctx, local, rank, world, log):
    """BASELINE configs[4]: encode + fader sweep + greedy decode; every rank runs an independent replica (no collective)."""
    from music_fader_nets_amd.synth import synth_batch
    from music_fader_nets_amd.decode import fader_sweep, greedy_decode
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)
    model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, H, Z, 32, n_component=K).to(dev)
    model.eval()
    b = synth_batch(np.random.RandomState(rank), DEC_SEQS, T, TR)
    d = torch.from_numpy(b["d"]).to(dev).to(torch.int32)
    c = torch.from_numpy(b["c"]).to(dev)
    torch.manual_seed(99 + rank)
    eps = (torch.randn(DEC_SEQS, Z).to(dev), torch.randn(DEC_SEQS, Z).to(dev))
    values = [-2.0 + 0.5 * k for k in range(DEC_VALUES)]
    rows = DEC_SEQS * DEC_VALUES
    for _ in range(max(1, args.warmup)):
        tok, _ = fader_sweep(model, d, c, values, steps=DEC_STEPS, which="r", eps=eps)
    torch.cuda.synchronize()
    if ctx is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        tok, _ = fader_sweep(model, d, c, values, steps=DEC_STEPS, which="r", eps=eps)
    torch.cuda.synchronize()
    if ctx is not None:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if ctx is not None:
        dt = _max_over_ranks(dt, dev)
    assert int(tok.min()) >= 0 and int(tok.max()) < 342
    tokens_per_s = world * rows * DEC_STEPS * args.steps / dt
    log("timed region done: %.3f ms per pass" % (dt / args.steps * 1e3))
    # the decode kernels alone (graph replay of 300 x 5 kernels), HIP events on the launch stream, mean of 5
    z = torch.randn(rows, 2 * Z + 24, device=dev)
    greedy_decode(model, z, DEC_STEPS, want_logp=False)
    ms = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        greedy_decode(model, z, DEC_STEPS, want_logp=False)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    ms = float(np.mean(ms))
    ach = rows * DEC_STEPS * F_ALG_DECODE_PER_TOKEN / (ms * 1e-3) / 1e12
    dops = model.engine().ops
    x6cells = bool(getattr(dops, "dw_x6", False) and getattr(dops, "cell_x6", False) and rows >= getattr(dops, "cell_x6_rows", 1 << 30) and rows % 128 == 0)
    dec_peak = PEAK_BF16X6_TFLOPS if x6cells else PEAK_F32_MFMA_TFLOPS      # the two cells are 93 % of a token's flops: the line is rated on their arithmetic
    dec_kernel = ("greedy decode of 2048 rows x 300 steps (graph of gru_cell_x6_kernel x 2 - the two cells on the bf16 MFMA with exact triple splits, producer / consumer form, layer 2 with "
                  "its input projection in the same K loop - and out_argmax_lds8_kernel<4> (fp32 MFMA) - output layer with the argmax in its epilogue - per token)") if x6cells else (
                  "greedy decode of 2048 rows x 300 steps (graph of gru_cell_wlds_ovl_kernel<4, 2, ...> x 2 - the two cells, weight slice in LDS "
                  "filled under the K loops, layer 2 with its input projection - and out_argmax_lds8_kernel<4> - output layer with the argmax in its epilogue - per token)")
    out = {
        "metric": "event-tokens/sec GM-VAE fader-sweep inference (encode + 8 fader values + 300-step greedy decode)",
        "value": round(tokens_per_s, 1), "unit": "event-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32/bf16x6" if x6cells else "f32",
        "data": "synthetic",
        "config": {"workload": "arousal-transfer / fader-sweep inference (BASELINE configs[4]): 256 sequences x T=256 encoded, 8 values "
                               "of z_r[:,0] each, 2048 rows x 300 greedy steps, hipGraph replay; replicas only for N>1",
                   "global_batch": rows * world, "seq_len": DEC_STEPS, "parallelism": "replicas%d" % world},
        "roofline": dict(bound="mfma", kernel=dec_kernel, achieved=round(ach, 2), peak=round(dec_peak, 1), arith="bf16x6" if x6cells else "f32",
                         frac_of_fp32_mfma_peak=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                         unit="TFLOP/s", frac=round(ach / dec_peak, 4), avg_launch_us=round(ms * 1e3, 1), traffic=None,
                         flop_per_launch=rows * DEC_STEPS * F_ALG_DECODE_PER_TOKEN,
                         weight_stream_GBs=round(DEC_STEPS * DECODE_WEIGHT_BYTES / (ms * 1e-3) / 1e9, 1)),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        out["cpu_baseline"] = cpu_baseline.time_decode_baseline(H, Z, rows, 30)      # the same 2048 rows as the GPU leg, 30 of its 300 steps
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--mode", choices=("train", "decode"), default="train")
    ap.add_argument("--sustain", type=int, default=200, help="train mode: more steps timed right behind the K timed ones -> sustained_ms_per_step (0 = skip)")
    ap.add_argument("--arith", choices=("f32", "bf16x6"), default=None,
                    help="arithmetic of the deep MFMA products (music-fader-nets_amd/arith.py; default: the package default).  Both are fp32-class: f32 = fp32 "
                         "MFMA chains, bf16x6 = exact bf16 triple splits, six partial products on the bf16 MFMA, fp32 accumulation")
    ap.add_argument("--no-other-arith", "--no-x6", dest="no_other_arith", action="store_true", help="train mode: skip the leg that times the OTHER arithmetic beside the headline")
    ap.add_argument("--leg-only", action="store_true", help=argparse.SUPPRESS)      # internal: the child process that times the other arithmetic
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="train mode: leave the configs[4] decode measurement out of the line")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args.gpus)

    # stdout carries exactly ONE line (the JSON of rank 0): libraries that chat on stdout (gloo's "[Gloo] Rank 0 is connected ...", RCCL's
    # version banner) are sent to stderr at the file-descriptor level, the result line is written to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    t_start = time.perf_counter()
    from mfn_import import load_package
    pkg = load_package()
    from music_fader_nets_amd import parallel

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ctx, local = parallel.init_from_env()       # control plane: gloo; the collectives of the step are RCCL calls through the C ABI
    world = 1 if ctx is None else ctx.world
    rank = 0 if ctx is None else ctx.rank
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(torch.device("cuda", local))

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    out = (bench_train if args.mode == "train" else bench_decode)(args, pkg, ctx, local, rank, world, log)
    if rank == 0:
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if ctx is not None:
        torch.distributed.barrier()
        if ctx.rccl is not None:
            ctx.rccl.close()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
