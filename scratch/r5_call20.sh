#!/bin/bash
# round 5: the one bit-identity failure of the 31-run soak (test_bursts_behind_the_last_bucket..., run 20 under --x6): how often, and on which arithmetic?
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_loop; rm -rf $out; mkdir -p $out
for mode in x6 f32; do
  fail=0
  for i in $(seq 1 40); do
    if [ $mode = x6 ]; then flag=--x6; else flag=; fi
    timeout 300 python -m pytest tests/test_parallel_rccl.py -x -q -m gpu $flag -k "bursts" -p no:cacheprovider > $out/${mode}_$i.log 2>&1
    rc=$?
    if [ $rc -ne 0 ]; then fail=$((fail+1)); else rm -f $out/${mode}_$i.log; fi
  done
  echo "$mode: $fail failures of 40" | tee -a $out/summary.txt
done
