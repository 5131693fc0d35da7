#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag5; rm -rf $out; mkdir -p $out
for i in 1 2; do
FN_EAGER_POOL=1 timeout 1500 python scratch/r5_bursts_diag.py f32 1500 none > $out/pool_$i.log 2>&1
grep "DIFFERS\|repetitions differ" $out/pool_$i.log
done
timeout 1500 python scratch/r5_bursts_diag.py f32 1500 none > $out/memset.log 2>&1
grep "DIFFERS\|repetitions differ" $out/memset.log
