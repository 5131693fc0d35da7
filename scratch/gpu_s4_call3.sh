#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/s4c3
cp scratch/lib_mt4all.so music-fader-nets_amd/libfadernets_hip.so
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "decode or greedy or eval or sweep or shift" > gpurun_out/s4c3/tests_mt4.log 2>&1; tail -3 gpurun_out/s4c3/tests_mt4.log
python scratch/bench_decode_mid.py scratch/lib_mt4all.so --one 2>&1 | grep -v amdgpu.ids > gpurun_out/s4c3/mt4all.txt
python scratch/bench_decode_mid.py scratch/lib_mt353.so --one 2>&1 | grep -v amdgpu.ids > gpurun_out/s4c3/mt448.txt
cp scratch/lib_mt353.so music-fader-nets_amd/libfadernets_hip.so
paste -d'\n' gpurun_out/s4c3/mt4all.txt gpurun_out/s4c3/mt448.txt | cut -c1-120
