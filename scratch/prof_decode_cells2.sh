#!/bin/bash
# rocprofv3 kernel stats of the tokens-only per-token decode (automatic kernel choice)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1536 800; do
  rm -rf /tmp/pd_$bi
  rocprofv3 --kernel-trace --stats -d /tmp/pd_$bi -o pd -- python $R/scratch/prof_decode_cells.py $bi > /tmp/pd_$bi.log 2>&1
  grep "Bi=" /tmp/pd_$bi.log
  echo "== $bi rows"; python $R/scratch/prof_summary.py /tmp/pd_$bi/pd_results.db 3 | cut -c1-200 | sed -n 2,6p
done
