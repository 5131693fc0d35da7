#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R
cp music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
for lib in "$@"; do
  cp $lib music-fader-nets_amd/libfadernets_hip.so
  echo "== $lib"
  timeout 300 python scratch/ab_x6.py 2>&1 | grep -v amdgpu.ids | grep -E "bf16x6=1|weight gradients=1" | head -4
done
cp /tmp/lib_ship.so music-fader-nets_amd/libfadernets_hip.so
