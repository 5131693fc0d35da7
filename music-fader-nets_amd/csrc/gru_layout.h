// gru_layout.h - private memory layouts shared by the per-step (gru.hip) and persistent (gru_persist.hip) GRU kernels.
#pragma once
#include "common.h"
#include "mma_core.h"

// offset of (batch row b, gate q in {r,z,n,hn}, hidden unit u) inside one step's gate slab; nrt = ceil(B/16).
// [unit tile][row tile][gate][i = b&3][quad = (b&15)>>2][j = u&15]: for fixed (tiles, gate, i) the 64 lanes
// (quad, j) of a wave touch 64 consecutive floats.
FN_DEVINL long gate_off(int b, int q, int u, int nrt) {
    return ((((long)(u >> 4) * nrt + (b >> 4)) * 4 + q) * 4 + (b & 3)) * 64 + ((b & 15) >> 2) * 16 + (u & 15);
}

// fragment-major image of a [rows][K] matrix (K % 32 == 0, rows padded to 16), NC = K / 32 chunks
FN_DEVINL long frag_off(int row, int k, int NC) {
    return ((((long)(row >> 4) * NC + (k >> 5)) * 2 + ((k >> 4) & 1)) * 64 + (((k & 15) >> 2) * 16 + (row & 15))) * 4 + (k & 3);
}

FN_DEVINL float f4at(const f32x4& v, int j) { return v[j]; }

// Gate arithmetic of ONE element of a GRU step (gmm_model.py:131-136 / nn.GRU), contraction pinned: every weight-stationary kernel that
// goes through here produces the same bits whatever the surrounding code looks like (the ping-pong kernels are tested bit for bit against
// the single-group ones).  x* = input-side pre-activations (b_ih + W_ih x ...), gh* = W_hh h + b_hh.
FN_DEVINL void fn_gru_gate(float xr, float xz, float xn, float ghr, float ghz, float ghn, float hprev, float& r, float& z, float& n, float& hnew) {
#pragma clang fp contract(off)
    r = fn_sigmoid(xr + ghr);
    z = fn_sigmoid(xz + ghz);
    n = fn_tanh(__builtin_fmaf(r, ghn, xn));
    hnew = __builtin_fmaf(z, hprev, (1.0f - z) * n);
}
// ... and of the backward step: dh = gradient wrt the new state -> gradients wrt the pre-activations (dr', dz', dn'), dn' r and the part of dh
// that flows straight into the previous state
FN_DEVINL void fn_gru_gate_bwd(float dh, float r, float z, float n, float hn, float hprev, float& dr, float& dz, float& dnp, float& dnr, float& carry) {
#pragma clang fp contract(off)
    const float dn = dh * (1.0f - z);
    const float dzz = dh * (hprev - n);
    dnp = dn * (1.0f - n * n);
    dz = dzz * z * (1.0f - z);
    dr = dnp * hn * r * (1.0f - r);
    dnr = dnp * r;
    carry = dh * z;
}

// ---- in-launch hand-over helpers of the weight-stationary kernels (gru_persist.hip, decode_persist.hip) ----------------------
// recipe R1 of cdna_hip_programming.md G16: 16-byte write-through (sc1) payload stores and L1-bypassing (sc1) loads; counters are
// relaxed agent-scope atomics.  The asm loads are not counted by hipcc: pair them with fn_wait_vm<N>() / fn_keep() (mma_core.h).
typedef unsigned int u32;
constexpr int RT = 256 + 16;                        // floats of one 16x16 accumulator tile in LDS: MFMA C layout + 4 floats per 16 lanes
constexpr u32 SPIN_LIMIT = 1u << 21;                // polls before a waiting workgroup gives up (~2 s)
FN_DEVINL void gld4_sc1(f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory"); }
// the trailing s_nop keeps hipcc from reusing the data registers before the store has read them
FN_DEVINL void stv4_sc1(float* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
FN_DEVINL f32x4 ldv4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
FN_DEVINL void stv4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
// orders every later use of v behind the preceding (volatile) wait: the value of an asm load must not be looked at before it
FN_DEVINL void fn_touch(f32x4& v) { asm volatile("" : "+v"(v)); }
FN_DEVINL u32 ld_cnt(u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// host-side entry points of gru_persist.hip: return FN_OK when the persistent kernel was launched,
// FN_PERSIST_NA when the configuration is not eligible (the caller then uses the per-step kernels).
#define FN_PERSIST_NA 1000000
int fn_gru_fwd_persist(const FnGruFwd* scans, int n_scans, hipStream_t st);
int fn_gru_bwd_persist(const FnGruBwd* scans, int n_scans, hipStream_t st);
int launch_pack(const float* src, int rows, int K, long ld, float* dst, hipStream_t st);
extern "C" int fn_frag3_pack(const float* src, int rows, int K, int ld, void* dst, void* stream);
