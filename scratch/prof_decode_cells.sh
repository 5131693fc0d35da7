#!/bin/bash
# per-token decode: FnGruCell.variant forced (8 staged, 4-7 LDS-free, 9-14 weights in LDS, 15-18 the same with the fills under the K loops, 0 automatic)
R=${GRAFT_REPO_ROOT:-/root/repo}
for bi in 2048 1536 1280 1024 800 640; do
  for v in 0 15 18 16 17; do
    python $R/scratch/prof_decode_cells.py $bi $v 2>&1 | grep "Bi="
  done
done
