#!/bin/bash
# round 4: timeline of one replayed step (same command as the kernel stats), decode lanes A/B, decode bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-x6 --no-cpu-baseline --no-decode > $O/prof_bench.log 2>&1
python $R/scratch/prof_summary.py $O/prof/bench_results.db 45 > $O/kernel_stats.txt
python $R/scratch/prof_timeline.py $O/prof/bench_results.db 100 3 > $O/timeline.txt 2>&1
rm -rf $O/prof
cd $R
timeout 600 python scratch/ab_decode_lanes.py 2>&1 | grep -v amdgpu.ids > $O/decode_lanes.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "row_ranges or large_decode or configs4" 2>&1 | tail -3
timeout 600 python bench.py --mode decode --steps 3 --warmup 1 > $O/bench_decode.json 2> $O/bench_decode.err
cat $O/timeline.txt; cat $O/decode_lanes.txt; cut -c1-600 $O/bench_decode.json
