#!/bin/bash
# round 5: which gradient of the eager data-parallel step goes nondeterministic (1 failure in 40 on both arithmetics, r5_call20)?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag; rm -rf $out; mkdir -p $out
for cfg in "bf16x6 150 none" "bf16x6 150 joined" "f32 150 none"; do
  set -- $cfg
  timeout 900 python scratch/r5_bursts_diag.py $1 $2 $3 > $out/$1_$3.log 2>&1
  tail -25 $out/$1_$3.log
done
