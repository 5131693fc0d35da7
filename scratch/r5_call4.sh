#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
for rep in 1 2; do
for lib in "" lib_ru6.so lib_ru4.so; do
  FN_LIB=$lib timeout 300 python scratch/r5_bench_fwd_scans.py 2>&1 | grep -E "library|bf16x6 ping-pong|fp32" | grep -v "64 steps"
done; done > gpurun_out/r5/call4_ab.log 2>&1
cat gpurun_out/r5/call4_ab.log
