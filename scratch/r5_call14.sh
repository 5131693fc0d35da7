#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
AB_ARITH=bf16x6 AB_REPS=2 timeout 900 python scratch/ab_engine.py "" "splitk_big=21" "splitk_big=32" "chunk=64" "ops.bwd_x6=False" "splitk_big=32,chunk=64" > gpurun_out/r5/call14_ab.log 2>&1
tail -9 gpurun_out/r5/call14_ab.log
