#!/bin/bash
# round 4, call 1: baseline bench line, VALU-in-MFMA-shadow microbenchmark, in-kernel stamps of the shipped scans
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 ./scratch/mfma_fill > gpurun_out/r4_mfma_fill.txt 2>&1
timeout 400 python bench.py > gpurun_out/r4_bench0.json 2> gpurun_out/r4_bench0.err
cp music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
timeout 300 python scratch/timing_persist_bwd.py > gpurun_out/r4_stamps0.txt 2>&1
cp /tmp/lib_ship.so music-fader-nets_amd/libfadernets_hip.so
cat gpurun_out/r4_mfma_fill.txt; grep -v amdgpu.ids gpurun_out/r4_stamps0.txt; head -c 1500 gpurun_out/r4_bench0.json
