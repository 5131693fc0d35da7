#!/bin/bash
# round 6: does the number of hardware queues (GPU_MAX_HW_QUEUES, default 4) change the captured step?  8 stream lanes share them.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
for q in 4 8 16; do
  echo "GPU_MAX_HW_QUEUES=$q"
  GPU_MAX_HW_QUEUES=$q AB_ARITH=bf16x6 AB_REPS=1 timeout 600 python scratch/ab_engine.py "" "losses_early=False" 2>&1 | grep "ms/step"
done; done
