#!/bin/bash
# round 6, final tree: the full gpu suite 8 times in a row (flakiness / race check of the three-lane schedule and the persistent GEMM launches), then scratch/soak.py 3000
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/final_soak; mkdir -p $O
for i in $(seq 1 ${1:-8}); do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $O/run_$i.log)"
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $O/run_$i.log | head -5; fi
done | tee $O/summary.txt
timeout 1500 python scratch/soak.py ${2:-3000} > $O/soak_steps.txt 2>&1; tail -12 $O/soak_steps.txt
