#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cp scratch/lib_r3h.so music-fader-nets_amd/libfadernets_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan or stationary or chunk or gru" > gpurun_out/c5_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/c5_tests.log; tail -4 gpurun_out/c5_tests.log
for lib in scratch/lib_r3base.so scratch/lib_r3h.so scratch/lib_r3base.so scratch/lib_r3h.so; do
  echo "== $lib" >> gpurun_out/c5_ab.log
  timeout 300 python scratch/bench_scan_ab.py $lib >> gpurun_out/c5_ab.log 2>&1
done
for lib in scratch/lib_r3base.so scratch/lib_r3h.so scratch/lib_r3base.so scratch/lib_r3h.so; do
  echo "== $lib" >> gpurun_out/c5_ab.log
  timeout 300 python scratch/ab_step.py $lib 0 >> gpurun_out/c5_ab.log 2>&1
done
cp scratch/lib_r3h.so music-fader-nets_amd/libfadernets_hip.so
grep -v amdgpu.ids gpurun_out/c5_ab.log
