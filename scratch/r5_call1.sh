#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests -x -q -m gpu -k "adversarial or decided_once or is_live or bf16x6 or benchmark_config or fused_gradients or three_train or gradient_slices or parallel" > gpurun_out/r5/call1_new_tests.log 2>&1
echo "new tests rc=$?"; tail -15 gpurun_out/r5/call1_new_tests.log
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/r5/call1_full.log 2>&1
echo "full rc=$?"; tail -5 gpurun_out/r5/call1_full.log
scratch/r5_soak_x6.sh 1 4
