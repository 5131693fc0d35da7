#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
timeout 300 python scratch/pp_stamps.py bwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pp_stamps_bwd.txt
cp /tmp/lib_ship.so music-fader-nets_amd/libfadernets_hip.so
