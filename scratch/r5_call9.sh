#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
for st in 0; do
  for cfg in "4 4" "2 4" "4 9" "2 33"; do
    timeout 120 python scratch/r5_dbg_bwd.py $st $cfg 2>&1 | grep -E "stage|fault|Error|error|fragb" | head -8
  done
done > gpurun_out/r5/call9_dbg.log 2>&1
cat gpurun_out/r5/call9_dbg.log
bash scratch/r5_call8.sh
