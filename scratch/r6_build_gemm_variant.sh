#!/bin/bash
# usage: r6_build_gemm_variant.sh <name> [-DX6W_... ...]  -> scratch/lib_<name>.so (gemm.hip rebuilt with experiment macros; other objects from the product build)
name=$1; shift
cd "$(dirname "$0")/../music-fader-nets_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 "$@" -c gemm.hip -o /tmp/gemm_$name.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o /tmp/gemm_$name.o gru.o gru_persist.o decode_persist.o embed.o loss.o optim.o comm.o -ldl -o ../../scratch/lib_$name.so && echo built lib_$name.so
