#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r5/bursts_diag2; rm -rf $out; mkdir -p $out
for i in 1 2 3; do
  timeout 1200 python scratch/r5_bursts_diag.py f32 400 joined snap > $out/f32_joined_$i.log 2>&1
  grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids\|socket.cpp\|Gloo" $out/f32_joined_$i.log | tail -60
done
