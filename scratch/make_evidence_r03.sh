#!/bin/bash
# session-4 call 1: full gpu suite, bench line, kernel stats + timeline, PMC passes of the step's kernels (after the hand-placed K loops)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s4c1
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
bash $R/scratch/pmc_step.sh > $O/pmc.log 2>&1
tail -4 $O/tests.log; tail -3 $O/bench_train.err; cut -c1-400 $O/bench_train.json; tail -30 $O/timeline.txt
