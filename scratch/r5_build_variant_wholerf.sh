#!/bin/bash
# scratch/lib_wholerf.so: gru_persist.hip with -DFN_WHOLE_RF (gru_bwd_rs_kernel keeps its scan descriptor in SGPRs instead of re-loading it from the kernarg segment)
cd "$(dirname "$0")/../music-fader-nets_amd/csrc"
r=/tmp/var_wholerf; rm -rf $r; d=$r/m/csrc; mkdir -p $d $r/include; cp *.h *.hip $d/; cp ../../include/*.h $r/include/
(cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DFN_WHOLE_RF -c gru_persist.hip -o gp.o) || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC api.o gemm.o gru.o $d/gp.o decode_persist.o embed.o loss.o optim.o comm.o -ldl -o ../../scratch/lib_wholerf.so && echo built lib_wholerf.so
cd $d && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -mllvm -amdgpu-mfma-vgpr-form=1 -DFN_WHOLE_RF --cuda-device-only -S -I$r/include -o gp.s gru_persist.hip 2>/dev/null
awk '/^_ZN12_GLOBAL__N_117gru_bwd_rs_kernelILi1EEEvNS_5QArgsE:/,/\.Lfunc_end/' gp.s | grep -c "s_load"
grep -A12 "amdhsa_kernel _ZN12_GLOBAL__N_117gru_bwd_rs_kernelILi1" gp.s | grep "next_free\|accum"; grep "sgpr_spill_count\|vgpr_spill_count" gp.s | sort | uniq -c | head
