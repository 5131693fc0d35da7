#!/bin/bash
# round-3 evidence files (gpurun_out/profiles_r03/ -> copy into profiles/): bench lines, kernel stats + timeline of the step, CU contention, DP soak
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r03
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_train.json 2> $O/bench_train.err
FN_FORCE_DIST=1 python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-decode > $O/bench_train_rccl_1rank.json 2> $O/bench_train_rccl_1rank.err
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
python $R/scratch/contention.py > $O/cu_contention.txt 2>&1
python $R/scratch/soak_dp.py ${1:-150} > $O/dp_graph_soak.txt 2>&1
tail -3 $O/bench_train.err; cut -c1-300 $O/bench_train.json; tail -12 $O/timeline.txt; tail -8 $O/cu_contention.txt; tail -3 $O/dp_graph_soak.txt
