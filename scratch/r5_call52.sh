#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for lib in lib_base.so lib_wholerf.so lib_base.so lib_wholerf.so; do
  echo "== step $lib"
  FN_LIB=$lib AB_REPS=2 timeout 300 python scratch/ab_engine.py "" 2>&1 | grep -v "amdgpu.ids" | tail -1
done
