#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/r5
timeout 300 python scratch/r5_timing_x6bwd.py > gpurun_out/r5/call10_timing_bwd.log 2>&1
tail -30 gpurun_out/r5/call10_timing_bwd.log
