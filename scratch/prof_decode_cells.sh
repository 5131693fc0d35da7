#!/bin/bash
# staged-GEMM cells against the LDS-free cell variants in the greedy decode at 2048 / 1024 / 800 rows
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1536 1024 800; do
  for v in 0 4 5 6 7; do
    python $R/scratch/prof_decode_cells.py $bi $v 2>&1 | grep "Bi="
  done
done
