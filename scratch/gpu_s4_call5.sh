#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4c5
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm or dwhh or benchmark_config or slices" > gpurun_out/s4c5/tests.log 2>&1; tail -3 gpurun_out/s4c5/tests.log
for lib in scratch/lib_mt353.so scratch/lib_xk.so scratch/lib_mt353.so scratch/lib_xk.so; do
  echo "== $lib"; python scratch/ab_step.py $lib 0 2>&1 | grep -v amdgpu.ids
done
cp scratch/lib_xk.so music-fader-nets_amd/libfadernets_hip.so
bash scratch/pmc_step.sh > gpurun_out/s4c5/pmc.log 2>&1
grep -A1 "grid 196608" gpurun_out/pmc_step/FETCH_SIZE.txt gpurun_out/pmc_step/WRITE_SIZE.txt gpurun_out/pmc_step/GRBM_GUI_ACTIVE.txt
