#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag6; rm -rf $out; mkdir -p $out
NSTEP=24 timeout 1500 python scratch/r5_bursts_diag.py f32 600 graph > $out/graph.log 2>&1
grep "DIFFERS\|repetitions differ\|Error\|error" $out/graph.log | head -20
timeout 1500 python scratch/r5_bursts_diag.py f32 1500 eager > $out/eager.log 2>&1
grep "DIFFERS\|repetitions differ\|Error\|error" $out/eager.log | head -20
