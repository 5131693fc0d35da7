// mma_core.h - LDS-staged operand tiles + f32 MFMA micro-kernel shared by the GEMM and GRU kernels.
//
// gfx950 only.  Arithmetic is v_mfma_f32_16x16x4_f32 (f32 in / f32 accumulate: bit-for-bit an fmaf
// chain, 32-cycle issue per SIMD, needs >= 2 independent accumulators per wave to reach the issue rate).
//
// Operand fragment convention of the instruction: lane l supplies A[i = l&15][k = l>>4] and
// B[k = l>>4][j = l&15]; D[row = (l>>4)*4 + reg][col = l&15].
// We permute K inside every 8-wide block so that one lane's two k-values are adjacent:
//   MFMA s (s=0,1) of block kb uses k = 8*kb + 2*(l>>4) + s       (same permutation for A and B)
// which lets a K-contiguous LDS tile be read with ONE ds_read_b64 per two MFMAs.
//
// LDS layouts (conflict-free by construction, see MI355X_MICROARCH.md LDS table):
//   KC (source is K-contiguous): tile[ROWS][BK+4] ; fragment = ds_read_b64 at row i, word 8kb+2g.
//        64-bank b64 slots: slot = i*(BK+4)/2 + g = 2i + g (mod 32) for BK%64==28.. (BK=32: 18i+g) ->
//        distinct over the 32 lanes of a half-wave for BK = 16 or 32.
//   RC (source is row-contiguous, i.e. the operand is stored transposed): tile[BK][ROWS+8];
//        fragment = two ds_read_b32 at k = 8kb+2g+s, word i: banks i and i+16 for g even/odd.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FN_DEVINL __device__ __forceinline__

// Loads whose completion WE count (cdna_hip_programming.md 5.7): hipcc's own s_waitcnt placement drains vmcnt(0) at every
// loop back-edge, which serialises a register prefetch ring.  An asm load is invisible to that bookkeeping: the destination is
// only valid after our own fn_wait_vm<N>() (+ sched_barrier so that no MFMA is hoisted above the wait), and every pipelined
// loop must end with fn_wait_vm<0>() before the registers can be reused.
FN_DEVINL void fn_gld4_asm(f32x4& dst, const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
FN_DEVINL void fn_gld1_asm(float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(p) : "memory"); }
// A fake use: keeps an asm-loaded register allocated (not re-used for anything else) up to this point.  Every register a
// fn_gld4_asm targets must reach a fn_keep() placed AFTER the wait that covers it, otherwise hipcc may recycle a
// "dead" destination while the load is still in flight.
FN_DEVINL void fn_keep(const f32x4& v) { asm volatile("" ::"v"(v)); }
template <int N>
FN_DEVINL void fn_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// bf16 x 6 arithmetic: x rounded to its nearest bf16 value (8 significant bits; ties away from zero: the bit pattern plus half a bf16 ulp,
// truncated).  Whatever the rounding, x - fn_rn16(x) is exact in fp32, and two levels leave a remainder of <= 8 significant bits: hi + mid + lo == x
// with all three pieces exact bf16 values.  ROUNDED pieces (instead of truncated ones, whose remainders all carry the sign of x) make the three
// partial products the arithmetic drops (mid*lo, lo*mid, lo*lo) zero-mean - a truncation split biases every dot product by ~2^-24 sum |a||b|
// towards zero.  (|x| >= 2^128 (1 - 2^-9) would round to infinity; nothing on this path comes near.)
FN_DEVINL float fn_rn16(float x) { return __uint_as_float((__float_as_uint(x) + 0x8000u) & 0xffff0000u); }

FN_DEVINL bool fn_aligned16(const void* p, long ld) { return ((((uintptr_t)p) & 15) == 0) && ((ld & 3) == 0); }

// Row maps: local tile row r -> source row.  valid(r) says whether the row exists; clamped(r) is always a legal row
// (out-of-range rows are loaded from a legal address and zeroed by a select, so the fast path has NO branches).
struct RowsPlain {
    int row0, nrows;
    FN_DEVINL bool valid(int r) const { return row0 + r < nrows; }
    FN_DEVINL long clamped(int r) const { return (long)min(row0 + r, nrows - 1); }
    FN_DEVINL bool all_valid(int rows) const { return row0 + rows <= nrows; }
};

// ROWS x BK operand tile staged through registers into LDS by NT threads.
template <int ROWS, int BK, bool KC, int NT>
struct Stage {
    static constexpr int NV = ROWS * BK / 4;
    static constexpr int NPT = (NV + NT - 1) / NT;
    static constexpr int LDW = KC ? (BK + 4) : (ROWS + 8);
    static constexpr int WORDS = KC ? ROWS * LDW : BK * LDW;
    float4 r[NPT];
    unsigned zero_mask = 0;      // bit q: r[q] belongs to a row outside the matrix -> store() writes zeros.  (Zeroing at LOAD
                                 // time would consume the register right after the load and serialise the prefetch.)

    // KC : element (row, k) at src[rowmap(row)*ld + k]
    // RC : element (row, k) at src[k*ld + rowmap(row)]   (rows of one float4 are consecutive)
    // can_fast(): 16-byte aligned source, K a multiple of BK [and all tile rows inside the matrix for RC]: then every tile
    // of the K loop can use load_fast = unconditional global_load_dwordx4, all issued back to back, NO branch and NO
    // consumer of the loaded registers (the loop stays software-pipelined).  Otherwise load_checked (element-wise bounds).
    template <class RowMap>
    static FN_DEVINL bool can_fast(const float* src, long ld, const RowMap& rowmap, int K) {
        return fn_aligned16(src, ld) && (K % BK == 0) && (KC || rowmap.all_valid(ROWS));
    }

    template <class RowMap>
    FN_DEVINL void load_fast(const float* __restrict__ src, long ld, const RowMap& rowmap, int k0) {
        zero_mask = 0;
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const int i = min((int)threadIdx.x + q * NT, NV - 1);
            if (KC) {
                const int row = i / (BK / 4), c = (i % (BK / 4)) * 4;
                r[q] = *reinterpret_cast<const float4*>(src + rowmap.clamped(row) * ld + (k0 + c));
                if (!rowmap.valid(row)) zero_mask |= 1u << q;
            } else {
                const int k = i / (ROWS / 4), c = (i % (ROWS / 4)) * 4;
                r[q] = *reinterpret_cast<const float4*>(src + (long)(k0 + k) * ld + rowmap.clamped(c));
            }
        }
    }

    template <class RowMap>
    FN_DEVINL void load_checked(const float* __restrict__ src, long ld, const RowMap& rowmap, int k0, int K) {
        zero_mask = 0;
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const int i = threadIdx.x + q * NT;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((NV % NT == 0) || i < NV) {
                if (KC) {
                    const int row = i / (BK / 4), c = (i % (BK / 4)) * 4;
                    const int gk = k0 + c;
                    if (rowmap.valid(row) && gk < K) {
                        const float* p = src + rowmap.clamped(row) * ld + gk;
                        v.x = p[0];
                        if (gk + 1 < K) v.y = p[1];
                        if (gk + 2 < K) v.z = p[2];
                        if (gk + 3 < K) v.w = p[3];
                    }
                } else {
                    const int k = i / (ROWS / 4), c = (i % (ROWS / 4)) * 4;
                    const int gk = k0 + k;
                    if (gk < K && rowmap.valid(c)) {
                        const float* p = src + (long)gk * ld + rowmap.clamped(c);
                        v.x = p[0];
                        if (rowmap.valid(c + 1)) v.y = p[1];
                        if (rowmap.valid(c + 2)) v.z = p[2];
                        if (rowmap.valid(c + 3)) v.w = p[3];
                    }
                }
            }
            r[q] = v;
        }
    }

    FN_DEVINL void store(float* __restrict__ lds) const {
#pragma unroll
        for (int q = 0; q < NPT; ++q) {
            const int i = threadIdx.x + q * NT;
            if ((NV % NT == 0) || i < NV) {
                const float4 v = ((zero_mask >> q) & 1u) ? make_float4(0.f, 0.f, 0.f, 0.f) : r[q];
                if (KC) {
                    const int row = i / (BK / 4), c = (i % (BK / 4)) * 4;
                    *reinterpret_cast<float4*>(lds + row * LDW + c) = v;
                } else {
                    const int k = i / (ROWS / 4), c = (i % (ROWS / 4)) * 4;
                    *reinterpret_cast<float4*>(lds + k * LDW + c) = v;
                }
            }
        }
    }

    // fragment pair (s = 0,1) for the 16 tile rows starting at tr0, k-block kb
    static FN_DEVINL float2 frag(const float* __restrict__ lds, int tr0, int kb, int lane) {
        const int i = lane & 15, g = lane >> 4;
        if (KC) {
            return *reinterpret_cast<const float2*>(lds + (tr0 + i) * LDW + 8 * kb + 2 * g);
        } else {
            float2 f;
            f.x = lds[(8 * kb + 2 * g) * LDW + tr0 + i];
            f.y = lds[(8 * kb + 2 * g + 1) * LDW + tr0 + i];
            return f;
        }
    }
};

// One BK-deep slab of MFMAs for a wave computing TM x TN tiles of 16x16.
template <int TM, int TN, int BK, class SA, class SB, int BSTRIDE = 16>
FN_DEVINL void mma_slab(const float* __restrict__ ldsA, const float* __restrict__ ldsB, int arow0, int brow0, int lane,
                        f32x4 (&acc)[TM][TN]) {
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
        float2 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = SA::frag(ldsA, arow0 + 16 * m, kb, lane);
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = SB::frag(ldsB, brow0 + BSTRIDE * n, kb, lane);      // BSTRIDE: B-tile rows between a wave's column tiles
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].x, b[n].x, acc[m][n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m].y, b[n].y, acc[m][n], 0, 0, 0);
    }
}

// Software-pipelined K loop of the LDS-staged GEMM: D register-staged tiles are kept in flight ahead of the tile being
// multiplied, LDS is double buffered, one barrier per K tile.  The steady state has NO branch around the loads (hipcc falls
// back to s_waitcnt vmcnt(0) at control-flow merges, which would serialise the ring): tiles past the end are re-loads of
// the last tile (clamped index, at most D redundant tiles) that are never stored.
template <int D, int TM, int TN, int BK, class SA, class SB, int BSTRIDE = 16, class FA, class FB>
FN_DEVINL void fn_kloop(float* __restrict__ smem, int nk, const FA& loadA, const FB& loadB, int arow0, int brow0, int lane,
                        f32x4 (&acc)[TM][TN]) {
    constexpr int BUFW = SA::WORDS + SB::WORDS;   // buffer b: A at smem + b*BUFW, B right behind it
    if (nk <= 0) return;
    SA sa[D];
    SB sb[D];
    const int last = nk - 1;
#pragma unroll
    for (int s = 0; s < D; ++s) {
        loadA(min(s, last) * BK, sa[s]);
        loadB(min(s, last) * BK, sb[s]);
    }
    sa[0].store(smem);
    sb[0].store(smem + SA::WORDS);
    __syncthreads();
    const int nk_pad = (nk + D - 1) / D * D;
    for (int base = 0; base < nk_pad; base += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int kt = base + u;
            const int cur = kt & 1;
            loadA(min(kt + D, last) * BK, sa[u]);           // set u held tile kt (already in LDS): refill with tile kt + D
            loadB(min(kt + D, last) * BK, sb[u]);
            if (kt < nk) mma_slab<TM, TN, BK, SA, SB, BSTRIDE>(smem + cur * BUFW, smem + cur * BUFW + SA::WORDS, arow0, brow0, lane, acc);
            sa[(u + 1) % D].store(smem + (cur ^ 1) * BUFW);  // tile kt + 1 (or a harmless duplicate of the last tile)
            sb[(u + 1) % D].store(smem + (cur ^ 1) * BUFW + SA::WORDS);
            __syncthreads();
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Gate non-linearities on the hardware transcendental units: exp2 (v_exp_f32) + reciprocal (v_rcp_f32), ~1 ulp each.
// |error| <= ~2e-7 absolute, far below the fp32 tolerance of the parity tests, and ~10x fewer instructions than
// expf / tanhf / IEEE division - the gate epilogue was measured at 1.8 us of a 10 us step otherwise.
// __builtin_amdgcn_rcpf IS v_rcp_f32; __frcp_rn (what this used to call) is the correctly rounded 1/x, which hipcc expands into the
// v_div_scale / v_rcp / 4 fma / v_div_fmas / v_div_fixup sequence - 12 of them per (row, 4 units) item, ~1 k cycles of a 36 k-cycle step.
FN_DEVINL float fn_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
FN_DEVINL float fn_tanh(float x) {
    const float e = __expf(-2.0f * fabsf(x));            // in (0, 1]: no overflow
    const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
    return copysignf(t, x);
}

// XCD-aware block remap: consecutive virtual ids land on the same XCD (block b runs on XCD b % 8),
// so tiles that share weight rows share one L2.  Bijective for any n.
#ifndef FN_XCD_REMAP
#define FN_XCD_REMAP 1
#endif
FN_DEVINL int fn_xcd_remap(int b, int n) {
    if (!FN_XCD_REMAP) return b;
    const int q = n >> 3, r = n & 7, x = b & 7, s = b >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + s;
}
