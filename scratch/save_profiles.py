"""copy the evidence of scratch/make_evidence_r03.sh (gpurun_out/s4c1 + gpurun_out/pmc_step) into profiles/r03_* with their headers"""
import os, json, subprocess, re
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); O = R + "/gpurun_out/s4c1"
b = open(O + "/bench_train.json").read()
open(R + "/profiles/r03_bench_train.json", "w").write(b)
ms = json.loads(b)["ms_per_step"]
hdr = """rocprofv3 --kernel-trace --stats -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode   (MI355X, round 3, final state: hand-placed K loops of the scans,
full-register weight-gradient GEMMs on the side lane with one K range set per XCD, dz partial products, LDS-free NT GEMMs beside the decoder pipeline; scratch/make_evidence_r03.sh)
+ the 5 eager passes of bench.py's per-kernel roofline measurement; summary of the rocpd kernel table by scratch/prof_summary.py
bench line of the un-profiled run in the same gpurun call: "ms_per_step": %s (profiles/r03_bench_train.json); the round's earlier collections (26.6 ms: column sums /
zero arena only; 23.4 ms: before the LDS-free NT GEMMs) are in git history (697f01b, 5a0e4b1..)
Reading the per-shape block at the end: the 196608-thread launches of gemm_tn_kernel (dW_hh-shaped products) average ~990 us over ALL launches of this command because
the replayed steps issue the decoder-side ones on the side lane in front of the encoder backward scan, where they wait for its CUs (max 4.7 ms); alone - the 5 eager passes of
bench.py's per-kernel measurement, every lane on one stream - they take 723-760 us (= `roofline.avg_launch_us` of the bench line).
""" % ms
open(R + "/profiles/r03_kernel_stats_bench_3steps.txt", "w").write(hdr + open(O + "/kernel_stats.txt").read())
hdr2 = """timeline of ONE replayed training step (kernel runs >= 100 us; scratch/prof_timeline.py on the same rocprofv3 database; all streams are reported as stream 0 by this rocprofv3)
round 3, final state (see r03_kernel_stats_bench_3steps.txt).  Phases: encoder forward | heads, latent | decoder pipeline forward (10 launches, the <= 128-register projection GEMM
co-resident with each) | output head, losses, dX of the output layer | decoder pipeline backward (dX of the layer-2 input in front of each launch) | dz, latent block, heads |
encoder backward scan | weight-gradient GEMMs of both sides (main + side lane) with the token-segment sums beside them | clip + Adam, weight images
"""
open(R + "/profiles/r03_timeline_one_step.txt", "w").write(hdr2 + open(O + "/timeline.txt").read())
print("ms_per_step", ms)
