#!/bin/bash
# normal library + the FN_TIMING build used by scratch/timing_*.py (absolute paths: callable from anywhere)
set -e
C=/root/repo/music-fader-nets_amd/csrc
make -C $C 2>&1 | grep -E "error|warning:" || true
mkdir -p /tmp/tb
for f in api gemm gru gru_persist decode_persist embed loss optim comm; do
  if [ ! -f /tmp/tb/$f.o ] || [ $C/$f.hip -nt /tmp/tb/$f.o ] || [ $C/gru_layout.h -nt /tmp/tb/$f.o ] || [ $C/mma_core.h -nt /tmp/tb/$f.o ] || [ $C/kloop_asm.h -nt /tmp/tb/$f.o ]; then
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DFN_TIMING -I/root/repo/include -c $C/$f.hip -o /tmp/tb/$f.o 2>&1 | grep -E "error" && exit 1 || true
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/tb/*.o -o /root/repo/scratch/lib_timing.so
ls -la /root/repo/music-fader-nets_amd/libfadernets_hip.so /root/repo/scratch/lib_timing.so
