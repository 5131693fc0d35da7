#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ping_pong" > gpurun_out/r4_pp_tests.log 2>&1; tail -15 gpurun_out/r4_pp_tests.log
timeout 300 python scratch/pp_time.py fwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pp_time.txt
