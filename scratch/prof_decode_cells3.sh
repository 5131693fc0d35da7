#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in 0 1 2 4 3 7; do
  rm -rf /tmp/pd_x
  FN_OA_DBG=$d rocprofv3 --kernel-trace --stats -d /tmp/pd_x -o pd -- python $R/scratch/prof_decode_cells.py 2048 > /tmp/pd_x.log 2>&1
  echo "== dbg $d"; python $R/scratch/prof_summary.py /tmp/pd_x/pd_results.db 4 | grep "out_argmax" | cut -c1-160
done
