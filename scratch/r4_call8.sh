#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd $R
timeout 300 python scratch/dec_T_sweep.py 2>&1 | grep -v amdgpu.ids | tee $O/dec_T_sweep.txt
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-x6 --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 0 3 > $O/timeline_full.txt 2>&1
rm -rf $O/prof
wc -l $O/timeline_full.txt
