#!/bin/bash
# Regenerates the evidence files of profiles/ for the current code on the GPU box (writes gpurun_out/profiles_r02/, copy what is to be
# judged into profiles/).  Usage on the box:  bash scratch/make_profiles.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profiles_r02
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
$R/scratch/mfma_clock > $O/mfma_clock.txt 2>/dev/null
python $R/bench.py > $O/bench_train.json 2> $O/bench_train.err
python $R/bench.py --mode decode --steps 3 --warmup 1 > $O/bench_decode.json 2> $O/bench_decode.err
rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 150 3 > $O/timeline.txt
python $R/scratch/bench_gemm.py > $O/gemm.txt 2>&1
python $R/scratch/bench_eg.py > $O/eg.txt 2>&1
rm -rf $O/prof
tail -3 $O/bench_train.err; cat $O/bench_train.json | cut -c1-400
