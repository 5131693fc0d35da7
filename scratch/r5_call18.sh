#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 900 python -m pytest tests -x -q -m gpu -k "forward_scan_bf16x6 or decided_once or is_live or chunked_scan or benchmark_config_vs or fused_gradients" > gpurun_out/r5/call18_tests.log 2>&1
echo "tests rc=$?"; tail -5 gpurun_out/r5/call18_tests.log
timeout 300 python scratch/r5_bench_fwd_scans.py 2>&1 | grep -v "64 steps" | tail -6
timeout 300 python scratch/r5_timing_x6pp.py 2>&1 | grep -A3 "bf16x6 pp rep 2"
AB_ARITH=bf16x6 AB_REPS=2 timeout 600 python scratch/ab_engine.py "" 2>&1 | tail -1
