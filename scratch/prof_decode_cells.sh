#!/bin/bash
# staged-GEMM cells (8) against the LDS-free cell variants in the greedy decode (0 = automatic choice)
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1280 1024 800; do
  for v in 8 0 4 6 5 7; do
    python $R/scratch/prof_decode_cells.py $bi $v 2>&1 | grep "Bi="
  done
done
