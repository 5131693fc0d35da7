#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ping_pong or scan or stationary or chunk" > gpurun_out/r4_pp_tests.log 2>&1; tail -5 gpurun_out/r4_pp_tests.log
timeout 300 python scratch/pp_time.py ${1:-both} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pp_time.txt
cp music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
timeout 300 python scratch/pp_stamps.py bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pp_stamps_bwd.txt
cp /tmp/lib_ship.so music-fader-nets_amd/libfadernets_hip.so
