#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04; mkdir -p $O; cd $R
FUZZ_SECONDS=240 SEED=4 timeout 400 python scratch/fuzz_pp.py 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_pp.txt | tail -5
FUZZ_SECONDS=120 SEED=5 timeout 300 python scratch/fuzz_persist.py 2>&1 | grep -v amdgpu.ids | tee $O/fuzz_persist.txt | tail -3
timeout 600 python scratch/soak.py 4000 2>&1 | grep -v amdgpu.ids | tee $O/soak.txt | tail -4
