#!/bin/bash
# usage: r6_ab_step.sh <libA or ""> <libB or ""> [rounds]  - captured benchmark step (bf16x6), alternating libraries in ONE session (box-to-box spread is +-5 %)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in $(seq 1 ${3:-3}); do
  for l in "$1" "$2"; do
    printf "%-22s " "${l:-product}"; FN_LIB=$l python scratch/r6_bench_lib.py --steps 30 --warmup 5 --sustain 0 --no-cpu-baseline --no-decode --no-other-arith 2>&1 | grep "timed region" | sed 's/.*done: //'
  done
done
