"""copy the evidence of scratch/make_evidence_r04.sh (+ r4_call7.sh) (gpurun_out/r04) into profiles/r04_* with their headers"""
import os, json, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); O = R + "/gpurun_out/r04"
P = R + "/profiles/"
b = open(O + "/bench_train.json").read().strip().split("\n")[-1]
open(P + "r04_bench_train.json", "w").write(b + "\n")
d = json.loads(b)
hdr = """rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 2 --sustain 0 --no-x6 --no-cpu-baseline --no-decode   (MI355X, round 4: ping-pong forward scans
gru_fwd_pp_kernel<1> (encoder, 128-row groups) / <2> (decoder pipeline, 64-row groups), register-stationary backward scans gru_bwd_rs_kernel<2> (encoder, 16 groups of
64 rows x 16 slices of 32 columns) / <1> (decoder pipeline, 32-row groups); scratch/make_evidence_r04.sh) + the 5 eager passes of bench.py's per-kernel roofline measurement;
summary of the rocpd kernel table by scratch/prof_summary.py.  Bench line of the un-profiled run in the same gpurun call: "ms_per_step": %s, "sustained_ms_per_step": %s
(profiles/r04_bench_train.json).  Reading the table: calls / 5 passes+steps; the scan symbols' averages agree with `roofline_all` of the bench line (enc fwd %s us, enc bwd %s us,
decoder chunks %s / %s us there); gemm_tn_kernel's mean mixes launch shapes (dW_hh-shaped 48 tiles x 16 K ranges: ~0.77 ms alone; max 4.3 ms = a side-lane launch waiting for the
encoder backward scan's CUs) - `roofline_by_symbol` of the bench line is the merged figure per symbol.
""" % (d["ms_per_step"], d.get("sustained_ms_per_step"), d["roofline_all"]["enc_fwd_scan"]["avg_launch_us"], d["roofline_all"]["enc_bwd_scan"]["avg_launch_us"],
       d["roofline_all"]["dec_fwd_scan_chunk"]["avg_launch_us"], d["roofline_all"]["dec_bwd_scan_chunk"]["avg_launch_us"])
open(P + "r04_kernel_stats_bench_3steps.txt", "w").write(hdr + open(O + "/kernel_stats.txt").read())
if os.path.exists(O + "/timeline.txt") and "Traceback" not in open(O + "/timeline.txt").read():
    hdr2 = """timeline of ONE replayed training step (kernel runs >= 100 us; scratch/prof_timeline.py on the same rocprofv3 database as r04_kernel_stats_bench_3steps.txt; all streams are
reported as stream 0 by this rocprofv3).  Phases: encoder forward | heads, latent | decoder pipeline forward (10 launches, the <= 128-register projection GEMM co-resident) | output
head, losses, dX of the output layer | decoder pipeline backward (dX of the layer-2 input in front of each launch) | dz, latent block, heads | encoder backward scan |
weight-gradient GEMMs of both sides (main + side lane) with the token-segment sums beside them | clip + Adam, weight images
"""
    open(P + "r04_timeline_one_step.txt", "w").write(hdr2 + open(O + "/timeline.txt").read())
pm = O + "/pmc_step"
if os.path.isdir(pm):
    hdr3 = """rocprofv3 --kernel-trace --pmc <group> -- python scratch/pmc_step.py 2      (scratch/pmc_step_r04.sh, summary lines by scratch/pmc_derive.py; MI355X, round 4 kernels)
Two eager (no hipGraph) forward+backward passes of the training step at the benchmark shape (hidden 512, B=256, T=256, Tr=64); one rocprofv3 pass per counter group (FETCH_SIZE and
WRITE_SIZE do not fit one pass; no other tracing domain).  PMC collection serialises dispatches: these are the counters of each kernel running ALONE.  Values = average per dispatch
over the matching kernel name (scratch/pmc_avg.py): per template instance (= launch shape) and per SYMBOL (all instances merged: `gru_fwd_pp_kernel`, `gru_bwd_rs_kernel`,
`gemm_tn_kernel` - what bench.py's `roofline` / `roofline_by_symbol` carry as `traffic`); the last block picks the 48-tile x 16-K-range launches of gemm_tn_kernel by grid size.
Units: FETCH_SIZE / WRITE_SIZE in KB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md) -> HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.
GRBM_GUI_ACTIVE is summed over the 8 XCDs (/8 = cycles of the dispatch); SQ_VALU_MFMA_BUSY_CYCLES is summed over 1024 SIMDs -> MFMA busy = (MFMA_BUSY / 1024) / (GUI_ACTIVE / 8).

derived (per dispatch):
"""
    tail = """
  algorithmic HBM bytes (SURVEY 8d): forward 8 KiB, backward 16 KiB per sample-step.
    encoder forward  gru_fwd_pp_kernel<1>: 262144 sample-steps -> 2.15 GB; measured 4.31 GB = 2.0 x (saved gates r, z, n, W_hn h: 4 H floats per sample-step on top of h)
    encoder backward gru_bwd_rs_kernel<2>: 4.29 GB; measured 8.04 GB = 1.87 x (write-through exchange slabs: 3 H floats per sample-step, saved gates read back)
    decoder pipeline launch (2 x 256 rows x 32 steps = 16384 sample-steps): forward 0.134 GB -> 0.313 GB measured (2.3 x), backward 0.268 GB -> 0.585 GB (2.2 x)
    per SYMBOL and average launch of a step (1 encoder + 10 pipeline launches): gru_fwd_pp_kernel 425984 x 8 KiB / 11 = 0.317 GB -> 0.676 GB measured;
      gru_bwd_rs_kernel 425984 x 16 KiB / 11 = 0.634 GB -> 1.262 GB measured (2.0 x; 2.0 TB/s over the symbol's 6.8 ms per step: not binding)
  MFMA pipe busy, round 3 -> round 4: encoder forward 74.5 -> 80.7 %, encoder backward 68.2 -> 79.6 %, decoder-pipeline forward 60.9 -> 62.4 %, backward 52.2 -> 56.4 %

"""
    body = ""
    for g in ("FETCH_SIZE", "WRITE_SIZE", "SQ_WAVES_SQ_INSTS_MFMA_S", "GRBM_GUI_ACTIVE"):
        body += "==================== %s ====================\n" % g + open(os.path.join(pm, g + ".txt")).read() + "\n"
    open(P + "r04_pmc_training_step.txt", "w").write(hdr3 + open(pm + "/derived.txt").read() + tail + body)
if os.path.exists(O + "/tests.log"):
    open(P + "r04_gpu_tests.txt", "w").write("python -m pytest tests -x -q -m gpu --durations=15   (MI355X, scratch/make_evidence_r04.sh)\n" + open(O + "/tests.log").read())
for name, dst, h in (("decode_lanes.txt", "r04_decode_row_range_lanes.txt", "scratch/ab_decode_lanes.py: greedy decode of N rows x 300 steps on the staged-GEMM cells (fn_gru_cell_f32 x 2 + output GEMM + argmax per token, one hipGraph), the batch cut\ninto 1-4 row ranges that run on their own streams (Engine.decode_lanes); best of 3 replays, host clock around a synchronised replay\n"),
                     ("bench_decode.json", "r04_bench_decode.json", "")):
    if os.path.exists(O + "/" + name):
        open(P + dst, "w").write(h + open(O + "/" + name).read())
print("ms_per_step", d["ms_per_step"], "sustained", d.get("sustained_ms_per_step"))
