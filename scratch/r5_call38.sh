#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in lib_base.so lib_nt4.so lib_base.so lib_nt4.so; do
  echo "== $lib"
  FN_LIB=$lib timeout 300 python scratch/r5_bench_bwd_scans.py 2>&1 | grep -v "amdgpu.ids" | tail -6
done
for lib in lib_base.so lib_nt4.so lib_base.so lib_nt4.so; do
  echo "== step $lib"
  FN_LIB=$lib AB_REPS=2 timeout 300 python scratch/ab_engine.py "" 2>&1 | grep -v "amdgpu.ids" | tail -3
done
