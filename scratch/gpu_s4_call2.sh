#!/bin/bash
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/s4c2
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for spec in side:0 after:0; do
  rocprofv3 --kernel-trace -d $O/prof_$spec -o t -- python $R/scratch/ab_dw.py $spec > $O/log_$spec.txt 2>&1
  python $R/scratch/prof_timeline.py $O/prof_$spec/t_results.db 15 > $O/timeline_$spec.txt
  rm -rf $O/prof_$spec
done
tail -3 $O/log_*.txt
