#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag4; rm -rf $out; mkdir -p $out
timeout 2400 python scratch/r5_bursts_diag.py f32 1500 none snap > $out/f32_none.log 2>&1
grep "DIFFERS\|repetitions differ\|g_dgx1 \|sd_dgx_r \|g_dghn1 \|        t = " $out/f32_none.log | cut -c1-400
