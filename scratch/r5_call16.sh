#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -s -k "gemm_tn_bf16x6 or adversarial or gru_weight_gradient or column_view" 2>&1 | grep -E "adversarial 2|passed|failed|rror" > gpurun_out/r5/call16_tests.log
cat gpurun_out/r5/call16_tests.log
timeout 300 python scratch/r5_bench_dwhh.py 2>&1 | grep -E "16 K|32 K|42 K"
AB_ARITH=bf16x6 AB_REPS=2 timeout 600 python scratch/ab_engine.py "" 2>&1 | tail -2
