#!/bin/bash
# first GPU call of round 3: scan-kernel parity with the flag-in-data hand-over, then A/B against the round-2 library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "scan or stationary or sentinel or chunk" > gpurun_out/c1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/c1_tests.log
tail -5 gpurun_out/c1_tests.log
for lib in scratch/lib_r2.so scratch/lib_r3f.so scratch/lib_r3g.so scratch/lib_r2.so scratch/lib_r3f.so scratch/lib_r3g.so; do
  echo "== $lib" >> gpurun_out/c1_ab.log
  timeout 300 python scratch/bench_scan_ab.py $lib >> gpurun_out/c1_ab.log 2>&1
done
for lib in scratch/lib_r2.so scratch/lib_r3f.so scratch/lib_r3g.so scratch/lib_r2.so scratch/lib_r3f.so scratch/lib_r3g.so; do
  echo "== $lib" >> gpurun_out/c1_ab.log
  timeout 300 python scratch/ab_step.py $lib 0 >> gpurun_out/c1_ab.log 2>&1
done
cat gpurun_out/c1_ab.log
timeout 300 python scratch/timing_flag.py > gpurun_out/timing_flag.log 2>&1
cp scratch/lib_keep.so music-fader-nets_amd/libfadernets_hip.so
grep -A4 "rep 2" gpurun_out/timing_flag.log
