#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 600 python -m pytest tests -x -q -m gpu -k "backward_scan_bf16x6 or forward_scan_bf16x6 or weight_images or test_abi" > gpurun_out/r5/call8_tests.log 2>&1
echo "bwd x6 tests rc=$?"; tail -30 gpurun_out/r5/call8_tests.log
timeout 300 python scratch/r5_bench_bwd_scans.py > gpurun_out/r5/call8_bench_bwd.log 2>&1
echo "bench rc=$?"; tail -8 gpurun_out/r5/call8_bench_bwd.log
