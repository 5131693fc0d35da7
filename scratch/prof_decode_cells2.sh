#!/bin/bash
# rocprofv3 kernel stats of the per-token decode at 2048 rows with the fused output layer + argmax
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048; do
  FUSED=0 python $R/scratch/prof_decode_cells.py $bi 0 | grep Bi=
  FUSED=1 python $R/scratch/prof_decode_cells.py $bi 0 | grep Bi=
  rm -rf /tmp/pd_$bi
  rocprofv3 --kernel-trace --stats -d /tmp/pd_$bi -o pd -- python $R/scratch/prof_decode_cells.py $bi > /tmp/pd_$bi.log 2>&1
  echo "== $bi rows"; python $R/scratch/prof_summary.py /tmp/pd_$bi/pd_results.db 6 | cut -c1-200 | head -12
done
