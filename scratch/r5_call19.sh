#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for lib in "" lib_noearly.so; do
FN_LIB=$lib AB_ARITH=bf16x6 AB_REPS=1 timeout 600 python scratch/ab_engine.py "" "prepack_h0=False" 2>&1 | grep -E "library|ms/step"
done; done
