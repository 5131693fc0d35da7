#!/bin/bash
# round 6, after the schedule work (three lanes, packs on the aux lane): the DP timing lines of r6_soak.sh again, a 60-capture soak of the DP graph, and the bit-identity hunt
# (r5_bursts_diag.py: fresh trainers x 8 eager steps / 100 captured steps, gradients compared bit for bit with the first trainer's) on the new schedule
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/soak2; mkdir -p $O
B="python bench.py --steps 20 --warmup 5 --sustain 0 --no-cpu-baseline --no-decode --no-other-arith"
( $B 2>&1 | grep "timed region" | sed 's/^/plain step (one hipGraph):                      /'
  FN_FORCE_DIST=1 FN_DP_GRAPH=1 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 FN_DP_GRAPH=1 (captured):        /'
  FN_FORCE_DIST=1 FN_DP_GRAPH=0 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 FN_DP_GRAPH=0 (eager launches):  /'
  FN_FORCE_DIST=1 $B 2>&1 | grep "timed region" | sed 's/^/FN_FORCE_DIST=1 (default of a 1-rank group):     /' ) > $O/dp_step.txt 2>&1
cat $O/dp_step.txt
timeout 900 python scratch/soak_dp.py 60 > $O/dp_graph_soak.txt 2>&1; tail -2 $O/dp_graph_soak.txt
R=${REPS:-800}
for a in bf16x6 f32; do
  timeout 1500 python scratch/r5_bursts_diag.py $a $R eager > $O/eager_single_$a.txt 2>&1; tail -1 $O/eager_single_$a.txt
  timeout 1500 python scratch/r5_bursts_diag.py $a $R none > $O/eager_dp_$a.txt 2>&1; tail -1 $O/eager_dp_$a.txt
done
NSTEP=100 timeout 1500 python scratch/r5_bursts_diag.py bf16x6 ${GREPS:-301} graph > $O/captured_bf16x6.txt 2>&1; tail -1 $O/captured_bf16x6.txt
