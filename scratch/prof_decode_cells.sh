#!/bin/bash
# per-token decode: FnGruCell.variant forced (8 staged, 4-7 LDS-free, 9-14 weights in LDS, 0 automatic)
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1536 1408 1280 1152 1024; do
  for v in 0 13 14 10; do
    python $R/scratch/prof_decode_cells.py $bi $v 2>&1 | grep "Bi="
  done
done
