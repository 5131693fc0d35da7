#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -k "gemm_tn_bf16x6 or adversarial or gru_weight_gradient or column_view" > gpurun_out/r5/call13_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/call13_tests.log
timeout 300 python scratch/r5_bench_dwhh.py > gpurun_out/r5/call13_dwhh.log 2>&1; cat gpurun_out/r5/call13_dwhh.log | tail -16
