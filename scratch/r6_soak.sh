#!/bin/bash
# round 6: (1) the data-parallel step through RCCL with one rank, captured and eager, on the round-6 kernels; (2) the 150-capture soak of the DP graph;
# (3) the eager-step nondeterminism hunt of round 5 re-run with gru_bwd_rs_kernel claiming the whole register file and NO eager guard (r5_bursts_diag.py:
# REPS fresh trainers x 8 eager steps, gradients compared bit for bit with the first trainer's); (4) the same on captured steps (100 steps per capture)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/soak; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-decode --no-other-arith"
( $B 2>&1 | grep "timed region" | sed 's/^/plain step (one hipGraph):                      /'
  FN_FORCE_DIST=1 FN_DP_GRAPH=1 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 FN_DP_GRAPH=1 (captured):        /'
  FN_FORCE_DIST=1 FN_DP_GRAPH=0 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 FN_DP_GRAPH=0 (eager launches):  /'
  FN_FORCE_DIST=1 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 (default of a 1-rank group):     /' ) > $O/dp_step.txt 2>&1
cat $O/dp_step.txt
timeout 1200 python scratch/soak_dp.py 150 > $O/dp_graph_soak.txt 2>&1; tail -2 $O/dp_graph_soak.txt
R=${REPS:-3000}
for a in f32 bf16x6; do
  timeout 2400 python scratch/r5_bursts_diag.py $a $R eager > $O/eager_single_$a.txt 2>&1; tail -1 $O/eager_single_$a.txt
  timeout 2400 python scratch/r5_bursts_diag.py $a $R none > $O/eager_dp_$a.txt 2>&1; tail -1 $O/eager_dp_$a.txt
done
NSTEP=100 timeout 2400 python scratch/r5_bursts_diag.py bf16x6 ${GREPS:-1001} graph > $O/captured_bf16x6.txt 2>&1; tail -1 $O/captured_bf16x6.txt
NSTEP=100 timeout 1200 python scratch/r5_bursts_diag.py f32 ${GREPS2:-301} graph > $O/captured_f32.txt 2>&1; tail -1 $O/captured_f32.txt
