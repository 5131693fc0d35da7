#!/bin/bash
cd "$(dirname "$0")/../music-fader-nets_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DFN_DBG_STAGES -c gru_persist.hip -o /tmp/gru_persist_dbg.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o gemm.o gru.o /tmp/gru_persist_dbg.o decode_persist.o embed.o loss.o optim.o comm.o -ldl -o ../../scratch/lib_dbg.so && echo built
