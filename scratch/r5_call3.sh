#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -s -k "adversarial" 2>&1 | grep -E "adversarial 2|passed|failed|Error" > gpurun_out/r5/call3_adv.log
cat gpurun_out/r5/call3_adv.log
timeout 300 python scratch/r5_timing_x6pp.py > gpurun_out/r5/call3_timing.log 2>&1
echo "timing rc=$?"; tail -40 gpurun_out/r5/call3_timing.log
