#!/bin/bash
# round 6: the full gpu suite N times on the final library (flakiness / race check, as profiles/r05_x6_suite_soak.txt), then the CPU-baseline thread probe
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r6/suite_soak; mkdir -p $O
python scratch/r6_cpu_threads_probe.py > $O/cpu_threads.txt 2>&1; cat $O/cpu_threads.txt
for i in $(seq 1 ${1:-10}); do
  timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider -x > $O/run_$i.log 2>&1; rc=$?
  echo "run $i rc=$rc $(tail -1 $O/run_$i.log)"
  if [ $rc -ne 0 ]; then grep -E "^(FAILED|ERROR)|Error|assert" $O/run_$i.log | head -5; fi
done | tee $O/summary.txt
