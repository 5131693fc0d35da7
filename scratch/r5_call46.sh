#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash scratch/r5_call_final.sh
rm -rf gpurun_out/soak_x6_supp; mkdir -p gpurun_out/soak_x6_supp
for i in $(seq 1 10); do
  t0=$(date +%s)
  timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/soak_x6_supp/run_$i.log 2>&1; rc=$?
  t1=$(date +%s)
  echo "run $i rc=$rc $((t1-t0))s :: $(tail -1 gpurun_out/soak_x6_supp/run_$i.log)" >> gpurun_out/soak_x6_supp/summary.txt
  [ $rc -eq 0 ] && rm -f gpurun_out/soak_x6_supp/run_$i.log
done
cat gpurun_out/soak_x6_supp/summary.txt
