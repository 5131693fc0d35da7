#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -k "forward_scan_bf16x6 or adversarial or gemm_tn_bf16x6" > gpurun_out/r5/call2_tests.log 2>&1
echo "x6 kernel tests rc=$?"; tail -25 gpurun_out/r5/call2_tests.log
timeout 300 python scratch/r5_bench_fwd_scans.py > gpurun_out/r5/call2_bench_fwd.log 2>&1
echo "bench rc=$?"; cat gpurun_out/r5/call2_bench_fwd.log | tail -12
