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

// host-side entry points of gru_persist.hip: return FN_OK when the persistent kernel was launched,
// FN_PERSIST_NA when the configuration is not eligible (the caller then uses the per-step kernels).
#define FN_PERSIST_NA 1000000
int fn_gru_fwd_persist(const FnGruFwd* scans, int n_scans, hipStream_t st);
int fn_gru_bwd_persist(const FnGruBwd* scans, int n_scans, hipStream_t st);
int launch_pack(const float* src, int rows, int K, long ld, float* dst, hipStream_t st);
