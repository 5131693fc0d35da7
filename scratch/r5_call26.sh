#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag3; rm -rf $out; mkdir -p $out
for cfg in "f32 1500 none" "f32 1500 joined"; do
  set -- $cfg
  timeout 1500 python scratch/r5_bursts_diag.py $1 $2 $3 > $out/$1_$3.log 2>&1
  grep "DIFFERS\|repetitions differ" $out/$1_$3.log
done
