#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do for lib in "" lib_gemm_prev.so; do
FN_LIB=$lib timeout 300 python scratch/r5_bench_dwhh.py 2>&1 | grep -E "library|16 K.*bf16|32 K.*bf16|42 K.*bf16"
FN_LIB=$lib AB_ARITH=bf16x6 AB_REPS=1 timeout 600 python scratch/ab_engine.py "" 2>&1 | tail -1
done; done
