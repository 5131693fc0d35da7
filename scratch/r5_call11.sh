#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -k "backward_scan_bf16x6" > gpurun_out/r5/call11_tests.log 2>&1
echo "bwd x6 tests rc=$?"; tail -5 gpurun_out/r5/call11_tests.log
timeout 300 python scratch/r5_bench_bwd_scans.py 2>&1 | grep -v "64 steps" | tail -5
timeout 300 python scratch/r5_timing_x6bwd.py 2>&1 | grep -A4 "bf16x6 rep 2"
