#!/bin/bash
# usage: [KLOOP4_NT=1 ...] r5_build_variant4.sh <name>  -> scratch/lib_<name>.so (gru_persist.hip rebuilt against a kloop4_asm.h generated under the caller's environment)
name=$1; shift
cd "$(dirname "$0")/../music-fader-nets_amd/csrc"
r=/tmp/var_$name; rm -rf $r; d=$r/m/csrc; mkdir -p $d $r/include; cp *.h *.hip $d/; cp ../../include/*.h $r/include/
KLOOP_EXPERIMENT=1 python3 gen_kloop4.py $d/kloop4_asm.h "$@" || exit 1
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -c gru_persist.hip -o gp.o) || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o gemm.o gru.o $d/gp.o decode_persist.o embed.o loss.o optim.o comm.o -ldl -o ../../scratch/lib_$name.so && echo built lib_$name.so
