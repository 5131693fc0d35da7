#!/bin/bash
# round 6: timeline of one captured training step with EVERY kernel (no duration threshold) -> gpurun_out/r6/timeline_full.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_tl -o bench -- python $R/bench.py --steps 3 --warmup 2 --sustain 0 --no-other-arith --no-cpu-baseline --no-decode > $O/prof_tl.log 2>&1
python $R/scratch/prof_timeline.py $O/prof_tl/bench_results.db 0 3 > $O/timeline_full.txt
rm -rf $O/prof_tl
wc -l $O/timeline_full.txt
