#!/bin/bash
# round 5: repeated runs of the gpu suite with the bf16 x 6 arithmetic as the package default (pytest --x6), every run its own process, stdout + stderr
# kept for runs that fail (VERDICT r4 task 1a: the one abort "without a message" of round 4).  usage: r5_soak_x6.sh <first> <last> [extra env assignments]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/soak_x6
mkdir -p $out
first=$1; last=$2; shift 2
for i in $(seq $first $last); do
  t0=$(date +%s)
  env "$@" timeout 900 python -m pytest tests -x -q -m gpu --x6 -p no:cacheprovider > $out/run_$i.log 2>&1
  rc=$?
  t1=$(date +%s)
  echo "run $i rc=$rc $((t1-t0))s env=[$*] :: $(tail -1 $out/run_$i.log)" >> $out/summary.txt
  if [ $rc -ne 0 ]; then
    dmesg 2>/dev/null | tail -40 > $out/dmesg_$i.txt
  else
    tail -3 $out/run_$i.log > $out/run_$i.tail; rm -f $out/run_$i.log
  fi
done
cat $out/summary.txt
