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
PEAK_HBM_GBS = 8000.0


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
            self.records.append((name, a, k, e0, e1))
            return res
        return call

    def __setattr__(self, name, value):
        setattr(self._ops, name, value)


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


def _symbol(name, a, k):
    """op call -> (kernel symbol as rocprofv3 prints it without template arguments, bound, work of that launch) for EVERY launch of the heavy
    symbols, whatever its shape (roofline_by_symbol merges what `_classify` splits by launch shape)"""
    if name in ("gru_seq_fwd", "gru_seq_bwd"):
        work = sum(s["B"] * s["T"] for s in a[0]) * FLOP_PER_SAMPLE_STEP
        return ("gru_fwd_pp_kernel" if name == "gru_seq_fwd" else "gru_bwd_rs_kernel"), "mfma", work
    if name == "gru_dwhh":
        rows, Hh = a[2].shape
        return "gemm_tn_kernel", "mfma", 2.0 * rows * 3 * Hh * Hh
    if name == "gemm":
        A, Cm = a[0], a[2]
        M, N = Cm.shape
        Kk = A.shape[1] if k.get("a_k", True) else A.shape[0]
        if k.get("a_k", True) and k.get("b_k", True):
            sym = "gemm_nt_direct_kernel / gemm_kernel"
        else:
            sym = "gemm_kernel" if k.get("a_k", True) else "gemm_tn_kernel"
        return sym, "mfma", 2.0 * M * N * Kk
    if name == "out_head":
        h, W = a[0], a[1]
        return "out_head_kernel", "mfma", 2.0 * h.shape[0] * W.shape[0] * h.shape[1]
    if name == "embed_grad_sorted":
        return "eg_piece_kernel + eg_final_kernel", "hbm", float(sum(j["dgx"].numel() for j in a[1]) * 4)
    return None


SYMBOL_NOTE = {
    "gru_fwd_pp_kernel": "forward weight-stationary scans, ping-pong over two row halves (all launches: encoder 4 x 256 rows x 256 steps, decoder pipeline chunks, attribute decoders)",
    "gru_bwd_rs_kernel": "backward weight-stationary scans, W_hh^T slice half register-stationary (all launches: encoder, decoder pipeline chunks, attribute decoders)",
    "gemm_tn_kernel": "weight-gradient products dW = dY^T X (dW_hh of every scan via fn_gru_dwhh_f32, dW of the dense layers)",
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


# HBM bytes per launch from rocprofv3 PMC passes (2 x FETCH_SIZE with the gfx950 correction + WRITE_SIZE, separate passes): bench.py
# itself cannot run the profiler, so these are the committed measurements of the same launches
PMC_TRAFFIC = {"enc_fwd_scan": 4.309e9, "enc_bwd_scan": 8.038e9, "dec_fwd_scan_chunk": 0.313e9, "dec_bwd_scan_chunk": 0.585e9, "dwhh_gemm_tn": 0.711e9,
               "out_head": 0.287e9}
# symbol -> mean bytes per launch over ALL of its launches in one step (1 encoder + 10 decoder-pipeline launches for the scans): per launch like `achieved`
PMC_TRAFFIC_SYMBOL = {"gru_fwd_pp_kernel": 0.676e9, "gru_bwd_rs_kernel": 1.262e9, "gemm_tn_kernel": 0.291e9}
PMC_SOURCE = "profiles/r04_pmc_training_step.txt (2 x FETCH_SIZE + WRITE_SIZE, separate rocprofv3 --pmc passes of the same launches)"


def roofline_rows(records):
    agg = {}
    for name, a, k, e0, e1 in records:
        c = _classify(name, a, k)
        if c is None or c[1] is None or c[0] not in ROW_INFO:
            continue
        ms = e0.elapsed_time(e1)
        ent = agg.setdefault(c[0], [0, 0.0, 0.0])
        ent[0] += 1
        ent[1] += ms
        ent[2] += c[1]
    rows = {}
    for row, (cnt, ms, work) in agg.items():
        bound, unit, kernel, share = ROW_INFO[row]
        if bound == "mfma":
            ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS * share, "TFLOP/s"
        else:
            ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS * share, "GB/s"
        rows[row] = dict(bound=bound, kernel=kernel, launches=cnt, avg_launch_us=round(ms / cnt * 1e3, 1), achieved=round(ach, 2),
                         peak=peak, unit=u, frac=round(ach / peak, 4), work_per_launch=work / cnt, total_us=ms * 1e3)
    return rows


def symbol_rows(records, reps):
    agg = {}
    for name, a, k, e0, e1 in records:
        c = _symbol(name, a, k)
        if c is None:
            continue
        ent = agg.setdefault(c[0], [c[1], 0, 0.0, 0.0])
        ent[1] += 1
        ent[2] += e0.elapsed_time(e1)
        ent[3] += c[2]
    rows = {}
    for sym, (bound, cnt, ms, work) in agg.items():
        if bound == "mfma":
            ach, peak, u = work / (ms * 1e-3) / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
        else:
            ach, peak, u = work / (ms * 1e-3) / 1e9, PEAK_HBM_GBS, "GB/s"
        rows[sym] = dict(bound=bound, kernel=sym, note=SYMBOL_NOTE.get(sym), launches_per_step=round(cnt / reps, 1), avg_launch_us=round(ms / cnt * 1e3, 1),
                         achieved=round(ach, 2), peak=peak, unit=u, frac=round(ach / peak, 4), work_per_launch=work / cnt,
                         us_per_step=round(ms * 1e3 / reps, 1))
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
    return dict(bound="mfma", rows=[r for r in SCAN_ROWS if r in rows], us_per_step=round(us, 1), achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS,
                unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), flop_per_step=work)


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


def bench_train(args, pkg, ctx, local, rank, world, log):
    from music_fader_nets_amd.synth import synth_batch
    dev = torch.device("cuda", local)
    torch.manual_seed(1234)                                 # identical weights on every rank
    model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, H, Z, 32, n_component=K).to(dev)
    trainer = pkg.GMVAETrainer(model, lr=1e-3, beta=0.2, dist_ctx=ctx)
    b = synth_batch(np.random.RandomState(rank), B, T, TR)  # each rank its own shard of the global batch (rank 0: the c1 fixture's batch)
    batch = trainer.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
    torch.manual_seed(99 + rank)
    eps = trainer.draw_eps(B, T)                            # the reference's draw order; the same eps every step
    log("model + batch ready on %s" % dev)
    if getattr(args, "x6_only", False):                     # child process of the default run: only the opt-in leg, its dict as the line
        del trainer
        return bench_x6_leg(args, pkg, batch, eps, dev, None, log)
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
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
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
        if rank == 0:
            out["comm"] = comm
    if rank == 0:
        # `roofline` = the kernel SYMBOL (template instances merged, as rocprofv3 --stats lists them) the step spends most of its time in;
        # `roofline_all` keeps the split by launch shape, `roofline_scans` the four scan rows as one time-weighted figure
        scans = scans_row(rows)
        for r in rows.values():
            r.pop("_reps", None)
        dom_sym = max(by_symbol, key=lambda r: by_symbol[r]["us_per_step"])
        dom = dict(by_symbol[dom_sym])
        dom.update(symbol=dom_sym, traffic=PMC_TRAFFIC_SYMBOL.get(dom_sym), traffic_source=PMC_SOURCE if dom_sym in PMC_TRAFFIC_SYMBOL else None,
                   flop_per_launch=dom.pop("work_per_launch"),
                   step_frac=round(tokens_per_s / world * F_ALG_PER_TOKEN / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))
        out["roofline"] = dom
        for r in by_symbol.values():
            r.pop("work_per_launch", None)
        out["roofline_by_symbol"] = by_symbol
        out["roofline_scans"] = scans
        for row, tr_ in PMC_TRAFFIC.items():
            if row in rows:
                rows[row]["traffic"], rows[row]["traffic_source"] = tr_, PMC_SOURCE
        out["roofline_all"] = rows
        out["roofline_worst"] = min(rows, key=lambda r: rows[r]["frac"])
        if first is not None and world == 1:                # same seeds as tests/golden/c1.npz (the reference's own train() at this size)
            out["first_step_loss"] = round(first[0], 4)
            gpath = os.path.join(ROOT, "tests", "golden", "c1.npz")
            if os.path.exists(gpath):
                ref = float(np.load(gpath)["train_tuples"][0][0])
                out["first_step_loss_reference"] = round(ref, 4)
                assert abs(first[0] - ref) <= 5e-4 * abs(ref), "first optimisation step deviates from the reference: %r vs %r" % (first[0], ref)
    if world == 1 and not args.no_x6:
        # OPT-IN arithmetic, reported BESIDE the headline (never inside it): the T*B-deep weight-gradient products and the forward scans on the
        # bf16 MFMA with every fp32 operand value cut exactly into three bf16 pieces (FN_GEMM_BF16X6, FnGruFwd.variant bit 14; HipOps.dw_x6) -
        # same seeds, same steps, its own trainer
        # in a CHILD process: whatever happens to that leg (it runs kernels outside the default path) cannot take the headline line down
        out["bf16x6_opt_in"] = _aux(lambda: x6_leg_in_child(args, first, log), log, "bf16x6 leg")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        out["cpu_baseline"] = _aux(lambda: cpu_baseline.time_baseline(H, Z, B, T, TR), log, "cpu baseline")
    if world == 1 and not args.no_decode:
        # BASELINE configs[4] rides along in the default line (about 0.3 s of GPU time): 1 warm-up + 3 timed passes
        del trainer
        dargs = argparse.Namespace(steps=3, warmup=1, no_cpu_baseline=args.no_cpu_baseline, sustain=0, no_x6=True)
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


def x6_leg_in_child(args, first, log):
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--x6-only", "--steps", str(args.steps), "--warmup", str(max(1, args.warmup)), "--sustain", "0",
           "--no-cpu-baseline", "--no-decode"]
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "FN_FORCE_DIST"):
        env.pop(k, None)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, env=env)
    for line in r.stderr.decode(errors="replace").splitlines():
        if line.startswith("[bench") and "bf16x6" in line:
            log("(child) " + line.split("] ", 1)[-1])
    lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
    if r.returncode != 0 or not lines:
        raise RuntimeError("child exited with %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:]))
    d = json.loads(lines[-1])
    if first is not None and "first_step_loss_fp32_path" in d:
        d["first_step_loss_fp32_path"] = round(first[0], 4)
    return d


def bench_x6_leg(args, pkg, batch, eps, dev, first, log):
    """the same timed region with HipOps.dw_x6 = True (weight-gradient GEMMs and forward scans: exact bf16 triple splits, 6 of 9 partial products,
    fp32 accumulation)"""
    torch.manual_seed(1234)
    model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, H, Z, 32, n_component=K).to(dev)
    trainer = pkg.GMVAETrainer(model, lr=1e-3, beta=0.2)
    model.engine().ops.dw_x6 = True                         # (after the trainer: it re-homes the parameters and with them the engine / its kernel table)
    model.weights_changed()                                 # the engine re-derives its weight images, now incl. the bf16 triple images of W_hh
    step, first6 = 20000, None
    for i in range(args.warmup):
        beta0, Bg = trainer.step_device(step, batch, eps)
        if i == 0:
            first6 = trainer._tuple8(beta0, Bg, False)
        step += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.step_device(step, batch, eps)
        step += 1
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    tup = trainer._tuple8(0.2, B, False)
    assert all(np.isfinite(tup)), tup
    log("bf16x6 (weight gradients + forward scans): %.3f ms/step" % (dt * 1e3))
    out = dict(ms_per_step=round(dt * 1e3, 3), value=round(B * T / dt, 1), unit="event-tokens/s", steps=args.steps, last_loss=round(tup[0], 4),
               first_step_loss=None if first6 is None else round(first6[0], 4),
               first_step_loss_fp32_path=None if first is None else round(first[0], 4),
               note="NOT the headline and not the default: fn_gru_dwhh_f32 / fn_gemm_f32(a_k=0, b_k=0) with FN_GEMM_BF16X6 and the forward weight-stationary scans "
                    "with FnGruFwd.variant bit 14 (gru_fwd_x6_kernel: weights and exchanged state as bf16 triples) - fp32 values cut EXACTLY into three bf16 pieces, "
                    "six of the nine exact partial products accumulated in fp32 on v_mfma_f32_16x16x32_bf16 (dropped terms <= 2^-24 |a b|); against float64 as "
                    "accurate as the fp32 MFMA kernels (tests/test_gpu_parity.py::test_gemm_tn_bf16x6, test_forward_scan_bf16x6, scratch/mfma_bf16x9.hip), the "
                    "benchmark shape passes the reference comparison at the same tolerances (test_benchmark_config_with_bf16x6_vs_reference_train) and the "
                    "whole gpu suite passes with the flag forced on (pytest -m gpu --x6); backward scans and every other kernel unchanged")
    del trainer, model
    return out


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
    out = {
        "metric": "event-tokens/sec GM-VAE fader-sweep inference (encode + 8 fader values + 300-step greedy decode)",
        "value": round(tokens_per_s, 1), "unit": "event-tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "arousal-transfer / fader-sweep inference (BASELINE configs[4]): 256 sequences x T=256 encoded, 8 values "
                               "of z_r[:,0] each, 2048 rows x 300 greedy steps, hipGraph replay; replicas only for N>1",
                   "global_batch": rows * world, "seq_len": DEC_STEPS, "parallelism": "replicas%d" % world},
        "roofline": dict(bound="mfma", kernel="greedy decode of 2048 rows x 300 steps (graph of gru_cell_direct_kernel x2 - layer 2 with its input projection - , "
                                              "out_argmax_lds_kernel (output layer with the argmax in its epilogue) per token)", achieved=round(ach, 2), peak=PEAK_F32_MFMA_TFLOPS,
                         unit="TFLOP/s", frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4), avg_launch_us=round(ms * 1e3, 1), traffic=None,
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
    ap.add_argument("--no-x6", action="store_true", help="train mode: skip the extra leg with the opt-in bf16 x 6 weight-gradient products")
    ap.add_argument("--x6-only", action="store_true", help=argparse.SUPPRESS)      # internal: the child process of the default run's opt-in leg
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
