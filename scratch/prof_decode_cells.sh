#!/bin/bash
# per-token decode: FnGruCell.variant forced (8 staged, 4-7 LDS-free, 9-12 weights in LDS, 0 automatic)
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1280 1024 800; do
  for v in 0 6 9 10 11 12; do
    python $R/scratch/prof_decode_cells.py $bi $v 2>&1 | grep "Bi="
  done
done
