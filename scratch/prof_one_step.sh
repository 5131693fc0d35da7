#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s4c8; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/prof -o t -- python $R/scratch/ab_dw.py side:0:lean_proj=1 > $O/log.txt 2>&1
python $R/scratch/prof_timeline.py $O/prof/t_results.db 100 > $O/timeline.txt
rm -rf $O/prof
tail -18 $O/timeline.txt
