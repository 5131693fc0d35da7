// x6w_core.h - shared pieces of the bf16 x 6 producer / consumer kernels (gemm.hip: gemm_tn_x6w_kernel, gemm_tn_x6v_kernel, gemm_nt_x6w_kernel;
// gru.hip: gru_cell_x6_kernel): the exact three-piece split of fp32 values, the LDS stage geometry, the barriers and the consumer wavefront.
// gfx950 only.  See gemm.hip for the design notes and the measurements behind them.
#pragma once
#include <type_traits>
#include <utility>

#include "mma_core.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
FN_DEVINL unsigned fn_pack_top16(float a, float b) {          // top 16 bits of a | top 16 bits of b << 16
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
FN_DEVINL float fn_top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
// element `e` of eight float4 vectors (8 consecutive k of one row) -> exact bf16 triple.  RN = rounded pieces (fn_rn16: one integer add more per
// level), else truncated ones.  One ROUNDED operand is enough to make the dropped partial products zero-mean (mid_a lo_b, lo_a mid_b: lo_b and
// mid_b then carry random signs); with both operands truncated they all have the sign of a b and bias a sum by ~2^-24 sum |a||b| towards zero
// (tests/test_gpu_parity.py::test_bf16x6_adversarial_operands_vs_float64).  This kernel is bound by these VALU operations: B rounded, A truncated.
// Two cheaper-on-paper forms measured SLOWER in one session (round 5, dW_hh product 1536 x 512 x 65280 at 16 / 32 K ranges; this form 698-736 / 603-610 us):
// v_cvt_pk_bf16_f32 for every piece (both halves of a dword rounded to nearest even by one instruction, 4.5 operations per value: 740 / 665 us) and
// two-element vector arithmetic that makes the remainders packed subtractions (v_pk_add_f32; 882 / 846 us: the compiler shuffles registers around them).
template <bool RN>
FN_DEVINL void fn_split8(const f32x4 (&v)[8], int e, bf16x8& h, bf16x8& m, bf16x8& l) {
    float x[8], hi[8], r1[8], mi[8], r2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = v[j][e];
#pragma unroll
    for (int j = 0; j < 8; ++j) { hi[j] = RN ? fn_rn16(x[j]) : fn_top16(x[j]); r1[j] = x[j] - hi[j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { mi[j] = RN ? fn_rn16(r1[j]) : fn_top16(r1[j]); r2[j] = r1[j] - mi[j]; }
    u32x4 H, M, L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        H[j] = fn_pack_top16(hi[2 * j], hi[2 * j + 1]);
        M[j] = fn_pack_top16(mi[2 * j], mi[2 * j + 1]);
        L[j] = fn_pack_top16(r2[2 * j], r2[2 * j + 1]);
    }
    h = __builtin_bit_cast(bf16x8, H);
    m = __builtin_bit_cast(bf16x8, M);
    l = __builtin_bit_cast(bf16x8, L);
}

constexpr int X6W_NT = 512;
// experiment switches of scratch/r6_build_gemm_variant.sh (never defined in the product build)
#ifndef X6W_PRIO_P
#define X6W_PRIO_P 0
#endif
#ifndef X6W_PRIO_C
#define X6W_PRIO_C 0
#endif
#ifndef X6W_SWAP
#define X6W_SWAP 0
#endif
#ifndef X6W_NS
#define X6W_NS 3                 // register sets of a producer wavefront (blocks in flight: NS - 1)
#endif
#ifndef X6W_INTERLEAVE
#define X6W_INTERLEAVE 1         // 0 (A/B builds): a block's loads as one burst in front of the cut
#endif
constexpr int X6W_SET = 4 * 3 * 64;              // u32x4 vectors of one set (64 operand columns x 32 k as triples: 12 KB)
constexpr int X6W_STAGE = 4 * X6W_SET;           // ... of one stage (48 KB)
constexpr int X6W_STAGES = 3;
// producers: every ds_write of the block has to be in LDS before the barrier; consumers: a bare s_barrier (their reads of the block the barrier
// retires were consumed by MFMAs long before; the reads in flight belong to the NEXT block, whose stage nobody writes for two more blocks)
FN_DEVINL void x6w_barrier_p() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}
FN_DEVINL void x6w_barrier_c() {
    asm volatile("s_barrier" ::: "memory");
}

template <int I>
using x6w_ic = std::integral_constant<int, I>;
template <int... I, class F>
FN_DEVINL void x6w_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(x6w_ic<I>{}), ...); }
template <int N, class F>
FN_DEVINL void x6w_for(F&& f) { x6w_for_impl(std::make_integer_sequence<int, N>{}, f); }      // f(integral_constant<0>) .. f(integral_constant<N - 1>), unrolled
FN_DEVINL f32x4 x6w_ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }

// Consumer wavefront of gemm_tn_x6w_kernel / gemm_nt_x6w_kernel: wave (wm, wn) multiplies sets wm (A) and 2 + wn (B) of every block into its 4 x 4
// accumulator tiles.  Operand registers: A triples of the current block and of the next one (two banks, block parity), B triples of output columns
// 0, 1 (first half of a block) and 2, 3 (second half).  While the first half of block t runs, B[2..3] of block t and A[0..1] of block t + 1 are
// read; during the second half A[2..3] and B[0..1] of block t + 1: no LDS latency is ever exposed, and no read of block t is in flight at the
// barrier that hands its stage back to the producers.
FN_DEVINL void x6w_consume(const u32x4* __restrict__ x6w_lds, int wm, int wn, int lane, int nblk, f32x4 (&acc)[4][4]) {
    bf16x8 Af[2][4][3], Bf[4][3];
    const u32x4* lA = x6w_lds + wm * X6W_SET + lane;
    const u32x4* lB = x6w_lds + (2 + wn) * X6W_SET + lane;
    auto rdA = [&](auto BANK, int stage, int a) __attribute__((always_inline)) {
        constexpr int bank = decltype(BANK)::value;
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Af[bank][a][pc] = __builtin_bit_cast(bf16x8, lA[stage * X6W_STAGE + (a * 3 + pc) * 64]);
    };
    auto rdB = [&](int stage, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bf[b][pc] = __builtin_bit_cast(bf16x8, lB[stage * X6W_STAGE + (b * 3 + pc) * 64]);
    };
    // the six products of a 32-k block, smallest first (piece 0 = hi, 1 = mid, 2 = lo): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    auto half = [&](auto BANK, auto B0) __attribute__((always_inline)) {      // 48 MFMAs: output columns b0, b0 + 1 of all four row tiles
        constexpr int bank = decltype(BANK)::value, b0 = decltype(B0)::value;
#ifdef X6W_EXP_NOMMA
        return;
#endif
#pragma unroll
        for (int ap = 0; ap < 2; ++ap)                   // row tiles 0, 1 first (their A triples were read half a block earlier than those of 2, 3)
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int a = 2 * ap; a < 2 * ap + 2; ++a)
#pragma unroll
                    for (int b = b0; b < b0 + 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[bank][a][PA[c]], Bf[b][PB[c]], acc[a][b], 0, 0, 0);
    };
    // block t (stage st), reading ahead in block t + 1 (stage st1).  The reads are unconditional - straight-line code, so that the compiler can
    // spread them between the MFMAs (sched_group_barrier) and count its LDS waits exactly; behind the last block they fetch a stage nobody uses
    auto step = [&](auto BANK, int st, int st1) __attribute__((always_inline)) {
        constexpr int bank = decltype(BANK)::value;
        const std::integral_constant<int, bank ^ 1> NB;
        rdB(st, 2);
        rdB(st, 3);
        rdA(NB, st1, 0);
        rdA(NB, st1, 1);
        half(BANK, std::integral_constant<int, 0>{});
#pragma unroll
        for (int q = 0; q < 12; ++q) {                   // 3 MFMAs : 1 read, the last 12 MFMAs of the half without (the reads land before the next half needs them)
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        rdB(st1, 0);
        rdB(st1, 1);
        rdA(NB, st1, 2);
        rdA(NB, st1, 3);
        half(BANK, std::integral_constant<int, 2>{});
#pragma unroll
        for (int q = 0; q < 12; ++q) {                   // 3 MFMAs : 1 read, the last 12 MFMAs of the half without (the reads land before the next half needs them)
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);
        x6w_barrier_c();
    };
    const std::integral_constant<int, 0> K0;
    const std::integral_constant<int, 1> K1;
    x6w_barrier_c();                                     // stages 0 and 1 hold blocks 0 and 1
    rdA(K0, 0, 0);                                       // (same order as the read-ahead of a step: the loop's wait counts hold for the first trip)
    rdA(K0, 0, 1);
    rdB(0, 0);
    rdB(0, 1);
    rdA(K0, 0, 2);
    rdA(K0, 0, 3);
    int st = 0;
#pragma unroll 1
    for (int t = 0; t < nblk; t += 2) {                  // one barrier per block, as many as the producers execute: 1 + nblk
        const int s1 = st == 2 ? 0 : st + 1, s2 = s1 == 2 ? 0 : s1 + 1;
        step(K0, st, s1);
        if (t + 1 < nblk) step(K1, s1, s2);
        st = s2;
    }
}

