#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5/prof15
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --arith bf16x6 --steps 3 --warmup 2 --sustain 0 --no-other-arith --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
head -60 $O/kernel_stats.txt; echo; tail -120 $O/timeline.txt | head -150
