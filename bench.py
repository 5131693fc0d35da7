#!/usr/bin/env python3
"""Headline benchmark: event-tokens/sec of one full MusicAttrRegGMVAE training step (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" = forward + every loss term + backward + gradient all-reduce (N>1) + clip + Adam on one synthetic
minibatch (trainer_gmm.py:220-258 semantics), inputs resident in HBM.  Workload: BASELINE config 1 - hidden 512,
z 128, K=2, B=256 sequences per GPU (weak scaling), T=256 event tokens, Tr=64 rhythm/note steps, fp32.
Prints ONE JSON line on rank 0 with the contract fields plus `roofline` and (N=1) `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H, Z, K, B, T, TR = 512, 128, 2, 256, 256, 64
# algorithmic work (SURVEY.md 8d / DESIGN.md): the recurrent part of one GRU step of one scan for one sample
FLOP_PER_SAMPLE_STEP = 2.0 * H * 3 * H                 # 1.573 MFLOP
F_ALG_PER_TOKEN = 37.12e6                              # whole training step, per event token
PEAK_F32_MFMA_TFLOPS = 157.3                           # MI355X_MICROARCH.md: v_mfma_f32_* dense peak


def measure_dominant_kernel(trainer, batch, eps, reps=3):
    """Launch duration of the dominant kernel, the weight-stationary encoder forward scan gru_fwd_persist_kernel<4,1,2,4>
    (ONE launch = T time steps x 4 scans x B rows), measured with HIP events on the stream it is launched on (torch's current
    stream).  The event pair also brackets the counter memset node that precedes the launch (a few microseconds of 4+ ms)."""
    eng = trainer.model.engine()
    d = batch[0]
    eng.encode(d)
    torch.cuda.synchronize()
    P = eng.p
    scans = []
    for e in ("r", "n"):
        for key, sfx, rev in ((e, "_l0", 0), (e + "_reverse", "_l0_reverse", 1)):
            pfx = "gru_%s." % e
            scans.append(dict(B=B, T=T, H=H, reverse=rev, w_hh_frag=eng.whh_f[key], b_hh=P[pfx + "bias_hh" + sfx],
                              b_ih=P[pfx + "bias_ih" + sfx], gx_table=eng.tab[key], idx=d, idx_shift=0,
                              h_all=eng.buf("enc_h_" + key, (T, B, H)), gates=eng.buf("enc_g_" + key, (T, eng.ops.gates_floats(B, H)))))
    best = None
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.ops.gru_seq_fwd(scans)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    if eng.ops.gru_sync_error():
        raise RuntimeError("weight-stationary scan: a workgroup gave up waiting (sync error flag set)")
    flop = T * 4 * B * FLOP_PER_SAMPLE_STEP
    achieved = flop / (best * 1e-3) / 1e12
    # traffic: HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, separate passes),
    # see profiles/r01_pmc_gru_fwd_persist_4scans.txt - bench.py itself cannot run the profiler, so this is the committed measurement.
    return dict(bound="mfma", kernel="gru_fwd_persist_kernel<4,1,2,4>", achieved=round(achieved, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                frac=round(achieved / PEAK_F32_MFMA_TFLOPS, 4), traffic=4.45e9, traffic_source="profiles/r01_pmc_gru_fwd_persist_4scans.txt",
                avg_launch_us=round(best * 1e3, 1), flop_per_launch=flop, steps_per_launch=T)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    args = ap.parse_args()

    t_start = time.perf_counter()
    from mfn_import import load_package
    pkg = load_package()
    from music_fader_nets_amd import parallel
    from music_fader_nets_amd.synth import synth_batch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    ctx, local = parallel.init_from_env("nccl")
    world = 1 if ctx is None else ctx.world
    rank = 0 if ctx is None else ctx.rank
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    torch.manual_seed(1234)                                 # identical weights on every rank
    model = pkg.MusicAttrRegGMVAE(342, 3, 16, 24, H, Z, 32, n_component=K).to(dev)
    trainer = pkg.GMVAETrainer(model, lr=1e-3, beta=0.2, dist_ctx=ctx)
    b = synth_batch(np.random.RandomState(rank), B, T, TR)  # each rank its own shard of the global batch
    batch = trainer.prepare_batch(b["d"], b["r"], b["n"], b["c"], b["r_density"], b["n_density"])
    torch.manual_seed(99 + rank)
    eps = (torch.randn(B, Z).to(dev), torch.randn(B, Z).to(dev))

    def log(msg):
        if rank == 0:
            print("[bench %.1fs] %s" % (time.perf_counter() - t_start, msg), file=sys.stderr, flush=True)

    log("model + batch ready on %s" % dev)
    step = 20000
    for _ in range(args.warmup):
        trainer.step_device(step, batch, eps)
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
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    tup = trainer._tuple8(0.2, B * world, False)            # one sync: the loss numbers of the last step (finite check)
    assert all(np.isfinite(tup)), tup

    tokens_per_s = world * B * T * args.steps / dt
    log("timed region done: %.3f ms/step (host enqueue %.3f ms/step)" % (dt / args.steps * 1e3, t_host / args.steps * 1e3))
    roof = measure_dominant_kernel(trainer, batch, eps)
    log("dominant-kernel timing done")
    roof["step_frac"] = round(tokens_per_s / world * F_ALG_PER_TOKEN / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)
    out = {
        "metric": "event-tokens/sec GM-VAE train, seq256 b256", "value": round(tokens_per_s, 1), "unit": "event-tokens/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "MusicAttrRegGMVAE train step (fwd+losses+bwd+clip+Adam), hidden 512, z 128, K=2, "
                               "B=256/GPU, T=256, Tr=64 (BASELINE configs[1]; N>1: DP, RCCL grad all-reduce)",
                   "global_batch": B * world, "seq_len": T, "parallelism": "dp%d" % world},
        "roofline": roof,
        "last_loss": round(tup[0], 4),
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_baseline
        out["cpu_baseline"] = cpu_baseline.time_baseline(H, Z, args.cpu_sample_batch, T, TR, steps=1)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if ctx is not None:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
