#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4_prof_pp -- python $GRAFT_REPO_ROOT/scratch/pp_time.py fwd > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r4_prof_pp -name "*kernel_stats*" | head -1 | xargs -I{} head -12 {}
cp music-fader-nets_amd/libfadernets_hip.so /tmp/lib_ship.so
timeout 300 python scratch/pp_stamps.py fwd 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4_pp_stamps_fwd.txt
cp /tmp/lib_ship.so music-fader-nets_amd/libfadernets_hip.so
