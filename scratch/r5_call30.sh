#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag7; rm -rf $out; mkdir -p $out
for i in 1 2; do
timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/eager_sc1_$i.log 2>&1
grep "DIFFERS\|repetitions differ\|Error\|error" $out/eager_sc1_$i.log | head -20
done
