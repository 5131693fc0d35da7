#!/bin/bash
# round 6, final evidence: PMC passes of the training step, gpu suite (package default = bf16x6, then --f32), default bench line (with the CPU baseline and the decode leg),
# decode bench, rocprofv3 kernel statistics + timeline of 3 captured steps.  Everything under gpurun_out/r6/final; the judged copies go to profiles/r06_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6/final
rm -rf $O; mkdir -p $O
cd $R
bash scratch/r6_pmc_step.sh > $O/pmc.log 2>&1; tail -40 $O/pmc.log | cut -c1-220
cp gpurun_out/pmc_r06/r06_pmc_traffic.json gpurun_out/pmc_r06/r06_pmc_training_step.txt $O/ 2>/dev/null
cp gpurun_out/pmc_r06/r06_pmc_traffic.json profiles/ 2>/dev/null          # what bench.py reads on this box (the committed copy is this file)
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $O/gpu_tests_default.log 2>&1; tail -2 $O/gpu_tests_default.log
timeout 900 python -m pytest tests -q -m gpu --f32 -p no:cacheprovider > $O/gpu_tests_f32.log 2>&1; tail -2 $O/gpu_tests_f32.log
timeout 900 python bench.py > $O/bench_train.json 2> $O/bench_train.err; tail -c 400 $O/bench_train.json; tail -4 $O/bench_train.err
timeout 600 python bench.py --mode decode > $O/bench_decode.json 2> $O/bench_decode.err; tail -c 300 $O/bench_decode.json
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-other-arith --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt
rm -rf $O/prof
head -14 $O/kernel_stats.txt | cut -c1-160
