// gemm.hip - dense fp32 GEMM on the f32 MFMA + small dense helpers (transpose, column sums, axpy, sum).
// Replaces nn.Linear forward / backward of the reference (gmm_model.py:86,91,108,113,123,137 and their
// autograd), see include/fadernets.h.
#include <atomic>
#include <type_traits>
#include <utility>

#include "common.h"
#include "mma_core.h"
#include "x6w_core.h"

namespace {

constexpr int NT = 256;
constexpr int PF_DEPTH = 2;     // K tiles in flight ahead of the MFMAs

// C tile BM x BN, 4 waves laid out WM x WN, each wave TM x TN MFMA tiles (16x16).
template <int BM, int BN, int BK, int WM, int WN, bool AKC, bool BKC>
__global__ __launch_bounds__(NT) void gemm_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                  const float* __restrict__ B, long ldb, float beta, float* __restrict__ C,
                                                  long ldc, const float* __restrict__ bias, int ksplit_len,
                                                  float* __restrict__ slabs) {
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16;
    using SA = Stage<BM, BK, AKC, NT>;
    using SB = Stage<BN, BK, BKC, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int ntn = (N + BN - 1) / BN, ntm = (M + BM - 1) / BM;
    const int tile = fn_xcd_remap(blockIdx.x, ntn * ntm);
    const int tm = tile / ntn, tn = tile % ntn;   // consecutive virtual tiles share the A row panel
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.z * ksplit_len;
    const int kend = min(K, kbeg + ksplit_len);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const RowsPlain ra{m0, M}, rb{n0, N};

    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (kend - kbeg + BK - 1) / BK;
    // one decision for the whole loop: the fast path needs aligned operands and a K range made of whole tiles
    const bool fast = SA::can_fast(A, lda, ra, kend - kbeg) && SB::can_fast(B, ldb, rb, kend - kbeg) && (kbeg % 4 == 0);
    if (fast) {
        auto loadA = [&](int k0, SA& st) { st.load_fast(A, lda, ra, kbeg + k0); };
        auto loadB = [&](int k0, SB& st) { st.load_fast(B, ldb, rb, kbeg + k0); };
        fn_kloop<PF_DEPTH, TM, TN, BK, SA, SB>(smem, nk, loadA, loadB, wm * TM * 16, wn * TN * 16, lane, acc);
    } else {
        auto loadA = [&](int k0, SA& st) { st.load_checked(A, lda, ra, kbeg + k0, kend); };
        auto loadB = [&](int k0, SB& st) { st.load_checked(B, ldb, rb, kbeg + k0, kend); };
        fn_kloop<PF_DEPTH, TM, TN, BK, SA, SB>(smem, nk, loadA, loadB, wm * TM * 16, wn * TN * 16, lane, acc);
    }

    // epilogue: D[row = (lane>>4)*4 + reg][col = lane&15]
    const int cj = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + (wn * TN + n) * 16 + cj;
            if (col >= N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + (wm * TM + m) * 16 + rq + i;
                if (row >= M) continue;
                const float v = acc[m][n][i];
                if (slabs) {
                    slabs[((long)blockIdx.z * M + row) * N + col] = v;
                } else {
                    float o = alpha * v;
                    if (bias) o += bias[col];
                    if (beta != 0.f) o += beta * C[(long)row * ldc + col];
                    C[(long)row * ldc + col] = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// Several small, independent GEMMs in ONE launch (blockIdx.z = job), each the sum of up to 4 (A_i, B_i) products over its own K range:
//     C_j = sum_i op(A_ji) op(B_ji) + beta_j C_j + bias_j
// The head / latent / dz phases of the step are chains of 256-row GEMMs that take a few microseconds each but a launch boundary apiece
// (8 launches for the four mu / var heads alone, each head = forward half + reverse half of the encoder state): 64 x 64 tiles.
constexpr int GM_MAX_JOBS = 12, GM_MAX_SEG = 4;
struct GmSeg {
    const float* A;
    const float* B;
    int lda, ldb, K;
};
struct GmJob {
    GmSeg seg[GM_MAX_SEG];
    float* C;
    const float* bias;
    int M, N, ldc, nseg;
    float beta;
};
struct GmArgs {
    GmJob job[GM_MAX_JOBS];
};

template <bool AKC, bool BKC>
__global__ __launch_bounds__(NT) void gemm_multi_kernel(const GmArgs args) {
    constexpr int BM = 64, BN = 64, BK = 16, WM = 2, WN = 2, TM = BM / WM / 16, TN = BN / WN / 16;
    using SA = Stage<BM, BK, AKC, NT>;
    using SB = Stage<BN, BK, BKC, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const GmJob& J = args.job[blockIdx.z];
    const int ntn = (J.N + BN - 1) / BN, ntm = (J.M + BM - 1) / BM;
    if ((int)blockIdx.x >= ntn * ntm) return;
    const int m0 = (blockIdx.x / ntn) * BM, n0 = (blockIdx.x % ntn) * BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const RowsPlain ra{m0, J.M}, rb{n0, J.N};
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int sI = 0; sI < J.nseg; ++sI) {
        const GmSeg& S = J.seg[sI];
        const float* A = S.A;
        const float* B = S.B;
        const long lda = S.lda, ldb = S.ldb;
        const int K = S.K, nk = (K + BK - 1) / BK;
        // These launches are a handful of workgroups walking K = 128 .. 2048 on the critical path between two scans.  Up to round 6 every segment
        // took the element-wise checked loads, whose selects consume a tile's registers right behind the loads: no prefetch at all, one exposed memory
        // latency per 16-k tile (1.2 us: 82 us for the encoder heads' 64 tiles).  Aligned segments of whole tiles (all of the step's) take the
        // unconditional 16-byte loads, 4 tiles in flight; the order of the summation is the same.
        if (SA::can_fast(A, lda, ra, K) && SB::can_fast(B, ldb, rb, K)) {
            auto loadA = [&](int k0, SA& st) { st.load_fast(A, lda, ra, k0); };
            auto loadB = [&](int k0, SB& st) { st.load_fast(B, ldb, rb, k0); };
            fn_kloop<4, TM, TN, BK, SA, SB>(smem, nk, loadA, loadB, wm * TM * 16, wn * TN * 16, lane, acc);
        } else {
            auto loadA = [&](int k0, SA& st) { st.load_checked(A, lda, ra, k0, K); };
            auto loadB = [&](int k0, SB& st) { st.load_checked(B, ldb, rb, k0, K); };
            fn_kloop<PF_DEPTH, TM, TN, BK, SA, SB>(smem, nk, loadA, loadB, wm * TM * 16, wn * TN * 16, lane, acc);
        }
    }
    const int cj = lane & 15, rq = (lane >> 4) * 4;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + (wn * TN + n) * 16 + cj;
            if (col >= J.N) continue;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + (wm * TM + m) * 16 + rq + i;
                if (row >= J.M) continue;
                float o = acc[m][n][i];
                if (J.bias) o += J.bias[col];
                if (J.beta != 0.f) o += J.beta * J.C[(long)row * J.ldc + col];
                J.C[(long)row * J.ldc + col] = o;
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// "TN" GEMM for the weight gradients  C[M][N] = sum_k A[k][m] B[k][n]  (both operands stored [K][rows], rows contiguous:
// dgx [T*B][3H], h [T*B][H], dlogits [T*B][344]).  No LDS and no barrier: with this storage the MFMA fragments can be
// read straight from global memory with full-width coalesced loads - lane (i = l&15, g = l>>4) loads the float4
// A[k0+g][m0+4i .. 4i+3], i.e. one operand value for each of FOUR interleaved 16-row MFMA tiles (tile a = rows m0+4i+a),
// 16 lanes = 256 contiguous bytes per k.  One wave owns a 64x64 output tile (4x4 MFMA tiles), 4 waves = 128x128 block,
// PF k-steps (4 k each) are kept in flight.  Output D[(l>>4)*4+r][l&15] of tile (a,b) is row m0+4((l>>4)*4+r)+a,
// col n0+4(l&15)+b -> the 4 b-values of a lane are one float4 store.
// ---------------------------------------------------------------------------------------------------------
FN_DEVINL float f4c(const f32x4& v, int j) { return v[j]; }

template <int PF>
FN_DEVINL void gemm_tn_body(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                            const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                            const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                            const float* __restrict__ A2, long lda2, int msplit) {
    const int ntn = (N + 127) / 128, ntm = (M + 127) / 128;
    // Split-K launches whose number of K ranges is a multiple of 8 come as a 1-D grid and give every XCD its own K ranges of ALL output
    // tiles (workgroup id % 8 = XCD): the tiles of one K range walk the same rows of A and B at the same time, so each XCD's L2 fetches
    // its share of the operands once.  With the tiles dealt to the XCDs instead (the 3-D grid below) every XCD streams all of B and a
    // sixth of A for every K range: 1.67 GB per dW_hh product against 0.54 GB of operands (PMC, profiles/r03_pmc_training_step.txt).
    int tile, zk;
    if (gridDim.z == 1 && slabs != nullptr) {
        const int S = (K + ksplit_len - 1) / ksplit_len, c = blockIdx.x & 7, q = blockIdx.x >> 3;
        zk = c * (S >> 3) + q / (ntn * ntm);
        tile = q % (ntn * ntm);
    } else {
        tile = fn_xcd_remap(blockIdx.x, ntn * ntm);
        zk = blockIdx.z;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (tile / ntn) * 128 + (wave >> 1) * 64, n0 = (tile % ntn) * 128 + (wave & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int kbeg = zk * ksplit_len, kend = min(K, kbeg + ksplit_len);
    // column offsets clamped inside the padded row (results of out-of-range rows/cols are never stored)
    // optional second source for the output rows >= msplit (a multiple of the 128-row tile): A = [A | A2] along M
    if (A2 != nullptr && m0 >= msplit) { A = A2 - msplit; lda = lda2; }
    // Column offsets of the 16-byte operand loads, kept inside the columns the operand REALLY has (M resp. N of them; results of
    // out-of-range rows / columns are never stored).  Clamping to the leading dimension instead is wrong for a column-offset view
    // (dpre[:, Z:2Z] of the latent block): its last row ends lda - offset floats behind the view's first column, and a load at lda - 4 ran
    // 128 bytes past the allocation - a memory fault once that allocation sat at the end of a mapped segment (found in round 4).
    const long mcols = A2 != nullptr ? (m0 >= msplit ? (long)M - msplit : (long)msplit) : (long)M;
    const long mrel = A2 != nullptr && m0 >= msplit ? (long)(m0 - msplit) : (long)m0;
    // (a lane's window of 4 columns is tied to its output columns: windows that still hold a valid column are read where they are - up to 3
    // floats of row padding, which ld % 4 == 0 guarantees - and only windows entirely beyond the last column move, to the last valid one)
    const long ca = (A2 != nullptr && m0 >= msplit ? msplit : 0) + min(mrel + 4 * li, ((mcols - 1) >> 2) << 2);
    const long cb = min((long)n0 + 4 * li, (((long)N - 1) >> 2) << 2);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m0 < M && n0 < N) {
        const int nks = (kend - kbeg + 3) >> 2;        // k-steps of 4 rows (the last one may be partial)
        const int nfull = (kend - kbeg) >> 2;           // steps whose 4 rows all exist
        const int nmain = nfull / PF * PF;               // software-pipelined part
        // Steady state rules (each violation was measured to serialise the ring): no branch around load(), nothing consumes the
        // loaded registers before their step, the MFMAs read fa[u]/fb[u] IN PLACE and load(u) refills the same registers right
        // after them (a copy would be rotated at the back-edge behind s_waitcnt vmcnt(0)), sched_barrier pins that order.
        if (nmain > 0) {
            f32x4 fa[PF], fb[PF];
            // running row pointers, one 64-bit add per load: the multiply-based address of every load (v_mul_lo_u32 / v_mad_u64_u32
            // are quarter-rate) sat between the last MFMA of a k-step and the first of the next one with the pipe draining
            const float* pa = A + (long)(kbeg + lg) * lda + ca;
            const float* pb = B + (long)(kbeg + lg) * ldb + cb;
            const long sa = 4 * lda, sb = 4 * ldb;
            auto load = [&](int set) {
                fn_gld4_asm(fa[set], pa);
                fn_gld4_asm(fb[set], pb);
                pa += sa;
                pb += sb;
            };
            auto mma = [&](int u) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(fa[u], a), f4c(fb[u], b), acc[a][b], 0, 0, 0);
            };
#pragma unroll
            for (int s = 0; s < PF; ++s) load(s);
            for (int base = 0; base + PF < nmain; base += PF) {   // steady state: every step refills its own ring slot
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    fn_wait_vm<2 * (PF - 1)>();          // the two loads of set u have landed; 2(PF-1) younger ones stay in flight
                    mma(u);
                    load(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fn_wait_vm<0>();                             // last PF steps: everything has been requested
#pragma unroll
            for (int u = 0; u < PF; ++u) mma(u);
#pragma unroll
            for (int s = 0; s < PF; ++s) { fn_keep(fa[s]); fn_keep(fb[s]); }
        }
        for (int ks = nmain; ks < nks; ++ks) {          // < PF + 1 leftover steps, unpipelined, K tail zeroed
            const int k = kbeg + 4 * ks + lg;
            const long kk = min(k, kend - 1);
            f32x4 va = *reinterpret_cast<const f32x4*>(A + kk * lda + ca);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(B + kk * ldb + cb);
            if (k >= kend) va = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(va, a), f4c(vb, b), acc[a][b], 0, 0, 0);
        }
    }
    const int colb = n0 + 4 * li;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (lg * 4 + r) + a;
            if (row >= M) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = colb + b;
                if (col >= N) continue;
                const float v = acc[a][b][r];
                if (slabs) {
                    slabs[((long)zk * M + row) * N + col] = v;
                } else {
                    float o = alpha * v;
                    if (bias) o += bias[col];
                    if (beta != 0.f) o += beta * C[(long)row * ldc + col];
                    C[(long)row * ldc + col] = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// The same "TN" product with EXACT fp32 products on the bf16 MFMA (v_mfma_f32_16x16x32_bf16: 16 x the fp32 MFMA rate) - "bf16 x 6":
//   every fp32 operand value is cut into three bf16 pieces, x = hi + mid + lo EXACTLY (B: rounded pieces, fn_rn16; A: truncated ones, hi = the top
//   16 bits of x, mid = the top 16 bits of x - hi, lo = the rest; each remainder is computed without rounding and the last one has <= 8 significant bits),
//   a b = sum_ij a_i b_j: nine partial products, each exact in the MFMA's fp32 accumulation.  The three smallest (lo x lo, lo x mid,
//   mid x lo: <= 2^-24 |a b|, below the rounding of the fp32 sum itself) are dropped; the other six are accumulated smallest first.
// Measured against float64 the result is as accurate as the fp32 MFMA chain (scratch/mfma_bf16x9.hip: max error 1.06e-7 vs 0.77e-7 of
// sum |a||b| at K = 4096, 1.37e-7 vs 1.91e-7 at K = 65536).  Chosen with FN_GEMM_BF16X6 (the package's default arithmetic "bf16x6" sets it, arith.py).
// THIS kernel (every wavefront splits its own operands) is the round-5 form, kept for A/B measurements and as the bit-identity reference of the
// producer / consumer kernel below (FN_GEMM_X6_PERWAVE).
// Layout: a 16x16x32 MFMA takes 8 consecutive k per lane, so lane (i, g) loads the float4 A[k0 + 8 g + j][m0 + 4 i ..] for j = 0..7 (the
// same 256-byte runs as the fp32 kernel, 8 loads per operand and 32-k block), element a of those eight vectors = the 8 k values of row
// 4 i + a of MFMA tile a; they are split in registers (5.5 VALU operations per value - what bounds this kernel: 1.38 x the fp32 rate
// with both operands split in the loop, 2.47 x with operands that arrive already split).  K tails (< 32) run on the fp32 MFMA.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void gemm_tn_x6_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                        const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                        const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                        const float* __restrict__ A2, long lda2, int msplit) {
    const int ntn = (N + 127) / 128, ntm = (M + 127) / 128;
    int tile, zk;
    if (gridDim.z == 1 && slabs != nullptr) {            // K ranges dealt to the XCDs (see gemm_tn_body)
        const int S = (K + ksplit_len - 1) / ksplit_len, c = blockIdx.x & 7, q = blockIdx.x >> 3;
        zk = c * (S >> 3) + q / (ntn * ntm);
        tile = q % (ntn * ntm);
    } else {
        tile = fn_xcd_remap(blockIdx.x, ntn * ntm);
        zk = blockIdx.z;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (tile / ntn) * 128 + (wave >> 1) * 64, n0 = (tile % ntn) * 128 + (wave & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    const int kbeg = zk * ksplit_len, kend = min(K, kbeg + ksplit_len);
    if (A2 != nullptr && m0 >= msplit) { A = A2 - msplit; lda = lda2; }
    // Column offsets of the 16-byte operand loads, kept inside the columns the operand REALLY has (M resp. N of them; results of
    // out-of-range rows / columns are never stored).  Clamping to the leading dimension instead is wrong for a column-offset view
    // (dpre[:, Z:2Z] of the latent block): its last row ends lda - offset floats behind the view's first column, and a load at lda - 4 ran
    // 128 bytes past the allocation - a memory fault once that allocation sat at the end of a mapped segment (found in round 4).
    const long mcols = A2 != nullptr ? (m0 >= msplit ? (long)M - msplit : (long)msplit) : (long)M;
    const long mrel = A2 != nullptr && m0 >= msplit ? (long)(m0 - msplit) : (long)m0;
    // (a lane's window of 4 columns is tied to its output columns: windows that still hold a valid column are read where they are - up to 3
    // floats of row padding, which ld % 4 == 0 guarantees - and only windows entirely beyond the last column move, to the last valid one)
    const long ca = (A2 != nullptr && m0 >= msplit ? msplit : 0) + min(mrel + 4 * li, ((mcols - 1) >> 2) << 2);
    const long cb = min((long)n0 + 4 * li, (((long)N - 1) >> 2) << 2);
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int kdone = kbeg;
    if (m0 < M && n0 < N) {
        const int nblk = (kend - kbeg) >> 5;             // whole 32-k blocks
        if (nblk > 0) {
            // two operand sets: while block t is split and multiplied out of one, the 16 loads of block t + 1 fill the other.  Plain loads
            // (the compiler counts vmcnt itself): under this kernel's register pressure it parks values in AGPRs, which the uncounted asm
            // loads of the fp32 kernel do not survive (their destinations are copied before the data has landed).
            f32x4 fa[2][8], fb[2][8];
            const float* pa = A + (long)(kbeg + 8 * lg) * lda + ca;
            const float* pb = B + (long)(kbeg + 8 * lg) * ldb + cb;
            const long sa = 32 * lda, sb = 32 * ldb;
            auto load = [&](int set) {
#pragma unroll
                for (int j = 0; j < 8; ++j) fa[set][j] = *reinterpret_cast<const f32x4*>(pa + j * lda);
#pragma unroll
                for (int j = 0; j < 8; ++j) fb[set][j] = *reinterpret_cast<const f32x4*>(pb + j * ldb);
                pa += sa;
                pb += sb;
            };
            auto mma = [&](int u) {
                bf16x8 bh[4], bm[4], bl[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) fn_split8<true>(fb[u], b, bh[b], bm[b], bl[b]);
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    bf16x8 ah, am, al;
                    fn_split8<false>(fa[u], a, ah, am, al);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm[b], acc[a][b], 0, 0, 0);
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh[b], acc[a][b], 0, 0, 0);
                }
            };
            load(0);
            int blk = 0;
#pragma unroll 1
            for (; blk + 2 <= nblk; blk += 2) {
                load(1);
                mma(0);
                if (blk + 2 < nblk) load(0);
                mma(1);
            }
            if (blk < nblk) mma(0);
            kdone = kbeg + 32 * nblk;
        }
        // K tail (< 32 rows): fp32 MFMA steps of 4 rows, unpipelined, last step zero-padded
        for (int k0 = kdone; k0 < kend; k0 += 4) {
            const int k = k0 + lg;
            const long kk = min(k, kend - 1);
            f32x4 va = *reinterpret_cast<const f32x4*>(A + kk * lda + ca);
            const f32x4 vb = *reinterpret_cast<const f32x4*>(B + kk * ldb + cb);
            if (k >= kend) va = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(va, a), f4c(vb, b), acc[a][b], 0, 0, 0);
        }
    }
    const int colb = n0 + 4 * li;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (lg * 4 + r) + a;
            if (row >= M) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = colb + b;
                if (col >= N) continue;
                const float v = acc[a][b][r];
                if (slabs) {
                    slabs[((long)zk * M + row) * N + col] = v;
                } else {
                    float o = alpha * v;
                    if (bias) o += bias[col];
                    if (beta != 0.f) o += beta * C[(long)row * ldc + col];
                    C[(long)row * ldc + col] = o;
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------
// The bf16 x 6 "TN" product with PRODUCER and CONSUMER wavefronts (round 6).  gemm_tn_x6_kernel above is bound by its VALU splits: every wave cuts its
// own 64 + 64 operand columns of a 32-k block into triples (~416 VALU instructions) beside 96 MFMAs, a wave's VALU work does not run in the shadow
// of its own MFMAs, and the two waves that share an operand both split it.  Sharing the split through LDS with four waves that all consume, then all
// split, then meet at a barrier was slower still (round 5, scratch/gemm_tn_x6_lds_shared_experiment.hip.txt): the phases run one after the other.
// Here a workgroup is EIGHT waves, two per SIMD: waves 0-3 only multiply (64 x 64 outputs each of the 128 x 128 tile: 24 ds_read_b128 + 96 MFMAs per
// block), waves 4-7 only produce (each loads one quarter of the block's operand values - 64 columns of A or of B x 32 k, 8 loads of 16 bytes per lane -
// cuts them into triples and writes them to LDS in MFMA operand order: ~208 VALU + 12 ds_write_b128).  The matrix pipe and the VALU are separate
// pipes of a SIMD: the producer's splits of block t + 1 run WHILE its SIMD neighbour multiplies block t.  Every operand value is split once per
// workgroup (half the VALU work of the kernel above) and loaded once (half the vector-memory traffic of the CU).  LDS: [stage 2][set 4][tile 4]
// [piece 3][64 lanes][8 bf16] = 96 KB, one workgroup per CU; ONE barrier per block (s_barrier behind lgkmcnt(0) only: the producers' global loads of
// block t + 2 stay in flight across it).  Same products in the same order per accumulator as gemm_tn_x6_kernel (lo*hi, hi*lo, mid*mid, mid*hi,
// hi*mid, hi*hi per 32 k; B pieces rounded, A pieces truncated): bit-identical results on K ranges of whole 32-k blocks; rows beyond the last row
// of the matrix count as zeros (the kernel above runs them on the fp32 MFMA).
// ---------------------------------------------------------------------------------------------------------
// Producer wavefront of gemm_tn_x6w_kernel: operand columns [pcol, pcol + 4) of rows kbeg + 32 blk + 8 lg + j (j = 0..7) of P, block after block,
// 8 loads per block into register set blk % NS.  In trip t the loads of block t + NS + 1 are requested, then block t + 2 (requested NS - 1 trips
// ago) is cut into LDS stage (t + 2) % 3, barrier; the consumers multiply block t meanwhile and read block t + 1 ahead.
// PLAIN loads, the compiler counts vmcnt: the steady-state trips are straight-line code (whole blocks only, no clamping, no branch between a request
// and its use), for which hipcc emits exactly `s_waitcnt vmcnt(8 (NS - 1))` in front of the cut (checked in the ISA; Makefile target isa.checked
// re-checks every build: csrc/check_isa_waits.py).  A first version with asm-statement loads and hand-counted waits passed every test, but its ISA
// showed whole register sets copied (`v_mov_b64`) at the control-flow merges of the partial-block / tail paths while asm loads into them could be in
// flight - the compiler cannot know: the hazard class of profiles/r05_x6_suite_soak.txt.  The last 2 NS + 1 blocks (the only ones that can be
// partial) run through guarded code with clamped row addresses and zero-filled rows.
template <bool RN>
FN_DEVINL void x6w_produce(u32x4* __restrict__ lds, const float* __restrict__ P, long pld, long pcol, int ps, int lane, int kbeg, int kend, int nblk) {
    constexpr int NS = X6W_NS;
    const int lg = lane >> 4;
    f32x4 fa[NS][8];
    auto gload = [&](auto SET, int blk) __attribute__((always_inline)) {              // a whole block inside the matrix
#ifndef X6W_EXP_NOLOAD
        constexpr int set = decltype(SET)::value;
        const float* p0 = P + ((long)kbeg + 32 * blk + 8 * lg) * pld + pcol;
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[set][j] = x6w_ld(p0 + j * pld);
#endif
    };
    auto gload_safe = [&](auto SET, int blk) __attribute__((always_inline)) {         // any block: rows beyond the matrix are read from its last row
#ifndef X6W_EXP_NOLOAD
        constexpr int set = decltype(SET)::value;
        const long k0 = (long)kbeg + 32 * blk + 8 * lg;
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[set][j] = x6w_ld(P + min(k0 + j, (long)kend - 1) * pld + pcol);
#endif
    };
    auto cut = [&](auto SET, int blk, int stage, bool zero_tail) __attribute__((always_inline)) {      // block blk: register set blk % NS -> LDS stage blk % 3
        constexpr int set = decltype(SET)::value;
        if (zero_tail) {                                 // rows beyond the matrix count as zeros (selects, no branch)
            const long k0 = (long)kbeg + 32 * blk + 8 * lg;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = k0 + j < kend;
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[set][j][e] = in ? fa[set][j][e] : 0.f;
            }
        }
        u32x4* dst = lds + stage * X6W_STAGE + ps * X6W_SET + lane;
#ifdef X6W_EXP_NOCUT
        if (blk > 1) return;
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8 h, m, l;
            fn_split8<RN>(fa[set], a, h, m, l);
            dst[(a * 3 + 0) * 64] = __builtin_bit_cast(u32x4, h);
            dst[(a * 3 + 1) * 64] = __builtin_bit_cast(u32x4, m);
            dst[(a * 3 + 2) * 64] = __builtin_bit_cast(u32x4, l);
        }
    };
    // prologue: blocks 0 and 1 cut, blocks 2 .. NS requested
    x6w_for<NS>([&](auto I) __attribute__((always_inline)) {
        if (decltype(I)::value < nblk) gload_safe(I, decltype(I)::value);
    });
    cut(x6w_ic<0>{}, 0, 0, true);
    if (nblk > 1) cut(x6w_ic<1>{}, 1, 1, true);
    if (nblk > NS) gload_safe(x6w_ic<0>{}, NS);
    x6w_barrier_p();
    int stage = 2, t = 0;
    // steady state, NS trips per pass (static register sets): trip t (t % NS == r) requests block t + NS + 1 -> set (r + 1) % NS and cuts block
    // t + 2 (set (r + 2) % NS, stage (t + 2) % 3).  Only whole blocks: t + NS + 1 <= nblk - 2 for every trip of the pass.
    // Request and cut are INTERLEAVED: two loads of the new block in front of every quarter of the cut.  Issued as one burst the 8 loads keep the
    // wave at the vector-memory queue until most of them have been taken (a CU's address path takes ~30 clocks per 1 KB wave load: measured, the
    // loads of a block cost the CU ~1000 clocks), and the address path then idles while the wave splits: loads + cut ran as long as their sum.
    auto fused = [&](auto LS, int lblk, auto CS, int stage_) __attribute__((always_inline)) {
        constexpr int ls = decltype(LS)::value, cs = decltype(CS)::value;
        const float* p0 = P + ((long)kbeg + 32 * lblk + 8 * lg) * pld + pcol;
        u32x4* dst = lds + stage_ * X6W_STAGE + ps * X6W_SET + lane;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef X6W_EXP_NOLOAD
            fa[ls][2 * a] = x6w_ld(p0 + (2 * a) * pld);
            fa[ls][2 * a + 1] = x6w_ld(p0 + (2 * a + 1) * pld);
#endif
            bf16x8 h, m, l;
            fn_split8<RN>(fa[cs], a, h, m, l);
            dst[(a * 3 + 0) * 64] = __builtin_bit_cast(u32x4, h);
            dst[(a * 3 + 1) * 64] = __builtin_bit_cast(u32x4, m);
            dst[(a * 3 + 2) * 64] = __builtin_bit_cast(u32x4, l);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (; t + 2 * NS + 1 < nblk; t += NS) {
        x6w_for<NS>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            if (X6W_INTERLEAVE) {
                fused(x6w_ic<(r + 1) % NS>{}, t + r + NS + 1, x6w_ic<(r + 2) % NS>{}, stage);
            } else {
                gload(x6w_ic<(r + 1) % NS>{}, t + r + NS + 1);
                cut(x6w_ic<(r + 2) % NS>{}, t + r + 2, stage, false);
            }
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        });
    }
    // the last <= 2 NS + 1 trips (t is a multiple of NS here: the register sets stay static)
    x6w_for<2 * NS + 1>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, r = i % NS;
        if (t + i < nblk) {
            if (t + i + NS + 1 < nblk) gload_safe(x6w_ic<(r + 1) % NS>{}, t + i + NS + 1);
            if (t + i + 2 < nblk) cut(x6w_ic<(r + 2) % NS>{}, t + i + 2, stage, true);
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        }
    });
}

// one (output tile, K range) of gemm_tn_x6w_kernel: both roles run 1 + nblk barriers, and behind the last one no stage is read any more (the reads in flight fetch
// values nobody uses): the next item of a workgroup that walks several of them may be cut into the stages at once
FN_DEVINL void gemm_tn_x6w_kernel_item(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                              const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                              const float* __restrict__ A2, long lda2, int msplit, u32x4* __restrict__ x6w_lds, int ntn, int tile, int zk) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mb = (tile / ntn) * 128, nb = (tile % ntn) * 128;          // the workgroup's output tile
    const int li = lane & 15, lg = lane >> 4;
    const int kbeg = zk * ksplit_len, kend = min(K, kbeg + ksplit_len);
    const int nblk = (kend - kbeg + 31) >> 5;
    if (nblk <= 0) return;                               // (the host never launches an empty K range)

    if (X6W_SWAP ? wave < 4 : wave >= 4) {
        if (X6W_PRIO_P) __builtin_amdgcn_s_setprio(X6W_PRIO_P);
        // ---- producer: set ps = 64 operand columns (sets 0, 1: A columns mb + 64 ps; sets 2, 3: B columns nb + 64 (ps - 2)) ----
        const int ps = wave & 3;
        const bool pa = ps < 2;
        const float* P = pa ? A : B;
        long pld = pa ? lda : ldb;
        // column offset of this lane's 16-byte loads, kept inside the columns the operand REALLY has (see gemm_tn_body)
        const int pc0 = pa ? mb + 64 * ps : nb + 64 * (ps - 2);
        long ncols = pa ? (long)M : (long)N, rel = pc0;
        if (pa && A2 != nullptr) {
            if (pc0 >= msplit) { P = A2; pld = lda2; ncols = (long)M - msplit; rel = pc0 - msplit; }
            else ncols = msplit;
        }
        long pcol = min(rel + 4 * li, ((ncols - 1) >> 2) << 2);
        if (rel >= ncols) pcol = ((ncols - 1) >> 2) << 2;                // a set entirely beyond the matrix: any legal column (never stored)
        if (pa) x6w_produce<false>(x6w_lds, P, pld, pcol, ps, lane, kbeg, kend, nblk);
        else x6w_produce<true>(x6w_lds, P, pld, pcol, ps, lane, kbeg, kend, nblk);
        return;
    }

    // ---- consumer ----
    if (X6W_PRIO_C) __builtin_amdgcn_s_setprio(X6W_PRIO_C);
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    x6w_consume(x6w_lds, wm, wn, lane, nblk, acc);
    const int m0 = mb + 64 * wm, n0 = nb + 64 * wn;
    const int colb = n0 + 4 * li;
    if (slabs != nullptr && (N & 3) == 0) {              // split-K slabs: 16-byte stores
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 4 * (lg * 4 + r) + a;
                if (row >= M || colb >= N) continue;
                const f32x4 v = {acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
                *reinterpret_cast<f32x4*>(slabs + ((long)zk * M + row) * N + colb) = v;
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (lg * 4 + r) + a;
            if (row >= M) continue;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int col = colb + b;
                if (col >= N) continue;
                const float v = acc[a][b][r];
                if (slabs) {
                    slabs[((long)zk * M + row) * N + col] = v;
                } else {
                    float o = alpha * v;
                    if (bias) o += bias[col];
                    if (beta != 0.f) o += beta * C[(long)row * ldc + col];
                    C[(long)row * ldc + col] = o;
                }
            }
        }
}

__global__ __launch_bounds__(X6W_NT) void gemm_tn_x6w_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                              const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                              const float* __restrict__ A2, long lda2, int msplit) {
    extern __shared__ __attribute__((aligned(16))) u32x4 x6w_lds[];      // [3 stages][4 sets][4 tiles][3 pieces][64 lanes]
    const int ntn = (N + 127) / 128, ntm = (M + 127) / 128;
    if (gridDim.z == 1 && slabs != nullptr) {            // K ranges dealt to the XCDs (see gemm_tn_body); a workgroup walks items blockIdx.x, + gridDim.x, ...
        const int S = (K + ksplit_len - 1) / ksplit_len, items = S * ntn * ntm;      // (gridDim.x < items only when it is a multiple of 8: the items of a workgroup stay on its XCD)
#pragma unroll 1
        for (int v = blockIdx.x; v < items; v += gridDim.x) {
            const int c = v & 7, q = v >> 3;
            gemm_tn_x6w_kernel_item(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit, x6w_lds, ntn, q % (ntn * ntm), c * (S >> 3) + q / (ntn * ntm));
        }
    } else {
        gemm_tn_x6w_kernel_item(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit, x6w_lds, ntn, fn_xcd_remap(blockIdx.x, ntn * ntm), blockIdx.z);
    }
}

// ---------------------------------------------------------------------------------------------------------
// The "NT" product C[M][N] = alpha sum_k A[m][k] B[n][k] (+ bias, + beta C) - nn.Linear forward / dX through a transposed weight image - on the
// bf16 MFMA with exact triple splits, producer / consumer form (round 6): the consumers are those of gemm_tn_x6w_kernel; a producer wave owns
// 64 rows of A or of B: lane (r = lane >> 2, q = lane & 3) loads k = 8 q .. 8 q + 7 (two 16-byte loads, 32 consecutive bytes) of row 4 r + e for
// e = 0..3 - the 8 consecutive k of MFMA operand lane (i = r, g = q) of tile e (tile e = rows 4 i + e: a consumer lane's four column tiles are four
// consecutive output columns, one 16-byte store).  Both operands K-contiguous with K % 32 == 0, M, N multiples of 128.  A: truncated pieces,
// B: rounded ones, six products smallest first, as everywhere.  Runs where the decoder pipeline turns layer-1 states into layer-2 gate inputs
// (gx2 = hx0 W_ih2^T per 32-step chunk) and gate gradients into state gradients (dhx0 = dgx2 W_ih2): 116 / 102 us per launch on the fp32 MFMA.
// ---------------------------------------------------------------------------------------------------------
template <bool RN>
FN_DEVINL void x6w_produce_nt(u32x4* __restrict__ lds, const float* __restrict__ P, long pld, int ps, int lane, int nblk) {
    constexpr int NS = X6W_NS;
    const int r = lane >> 2, q = lane & 3;
    f32x4 fa[NS][8];                                     // [set][2 e + j]: row 4 r + e, k = 8 q + 4 j ..
    const float* base = P + (long)(4 * r) * pld + 8 * q;
    auto gload = [&](auto SET, int blk) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const float* p0 = base + 32 * blk;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            fa[set][2 * e] = x6w_ld(p0 + e * pld);
            fa[set][2 * e + 1] = x6w_ld(p0 + e * pld + 4);
        }
    };
    auto cut = [&](auto SET, int stage) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        u32x4* dst = lds + stage * X6W_STAGE + ps * X6W_SET + r + 16 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x[8], hi[8], r1[8], mi[8], r2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = fa[set][2 * e + (j >> 2)][j & 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = RN ? fn_rn16(x[j]) : fn_top16(x[j]); r1[j] = x[j] - hi[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { mi[j] = RN ? fn_rn16(r1[j]) : fn_top16(r1[j]); r2[j] = r1[j] - mi[j]; }
            u32x4 Hh, Mm, Ll;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Hh[j] = fn_pack_top16(hi[2 * j], hi[2 * j + 1]);
                Mm[j] = fn_pack_top16(mi[2 * j], mi[2 * j + 1]);
                Ll[j] = fn_pack_top16(r2[2 * j], r2[2 * j + 1]);
            }
            dst[(e * 3 + 0) * 64] = Hh;
            dst[(e * 3 + 1) * 64] = Mm;
            dst[(e * 3 + 2) * 64] = Ll;
        }
    };
    // same trip structure as x6w_produce (whole blocks only here: K % 32 == 0): blocks 0, 1 cut and 2 .. NS requested, then per trip
    // request t + NS + 1 | cut t + 2 | barrier
    x6w_for<NS>([&](auto I) __attribute__((always_inline)) {
        if (decltype(I)::value < nblk) gload(I, decltype(I)::value);
    });
    cut(x6w_ic<0>{}, 0);
    if (nblk > 1) cut(x6w_ic<1>{}, 1);
    if (nblk > NS) gload(x6w_ic<0>{}, NS);
    x6w_barrier_p();
    int stage = 2, t = 0;
    // request and cut interleaved (see x6w_produce): the two loads of row 4 r + e of the new block in front of the cut of tile e
    auto fused = [&](auto LS, int lblk, auto CS, int stage_) __attribute__((always_inline)) {
        constexpr int ls = decltype(LS)::value, cs = decltype(CS)::value;
        const float* p0 = base + 32 * lblk;
        u32x4* dst = lds + stage_ * X6W_STAGE + ps * X6W_SET + r + 16 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            fa[ls][2 * e] = x6w_ld(p0 + e * pld);
            fa[ls][2 * e + 1] = x6w_ld(p0 + e * pld + 4);
            float x[8], hi[8], r1[8], mi[8], r2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = fa[cs][2 * e + (j >> 2)][j & 3];
#pragma unroll
            for (int j = 0; j < 8; ++j) { hi[j] = RN ? fn_rn16(x[j]) : fn_top16(x[j]); r1[j] = x[j] - hi[j]; }
#pragma unroll
            for (int j = 0; j < 8; ++j) { mi[j] = RN ? fn_rn16(r1[j]) : fn_top16(r1[j]); r2[j] = r1[j] - mi[j]; }
            u32x4 Hh, Mm, Ll;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Hh[j] = fn_pack_top16(hi[2 * j], hi[2 * j + 1]);
                Mm[j] = fn_pack_top16(mi[2 * j], mi[2 * j + 1]);
                Ll[j] = fn_pack_top16(r2[2 * j], r2[2 * j + 1]);
            }
            dst[(e * 3 + 0) * 64] = Hh;
            dst[(e * 3 + 1) * 64] = Mm;
            dst[(e * 3 + 2) * 64] = Ll;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (; t + 2 * NS < nblk; t += NS) {
        x6w_for<NS>([&](auto R) __attribute__((always_inline)) {
            constexpr int rr = decltype(R)::value;
            if (X6W_INTERLEAVE) {
                fused(x6w_ic<(rr + 1) % NS>{}, t + rr + NS + 1, x6w_ic<(rr + 2) % NS>{}, stage);
            } else {
                gload(x6w_ic<(rr + 1) % NS>{}, t + rr + NS + 1);
                cut(x6w_ic<(rr + 2) % NS>{}, stage);
            }
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        });
    }
    x6w_for<2 * NS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, rr = i % NS;
        if (t + i < nblk) {
            if (t + i + NS + 1 < nblk) gload(x6w_ic<(rr + 1) % NS>{}, t + i + NS + 1);
            if (t + i + 2 < nblk) cut(x6w_ic<(rr + 2) % NS>{}, stage);
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        }
    });
}

__global__ __launch_bounds__(X6W_NT) void gemm_nt_x6w_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                              const float* __restrict__ bias) {
    extern __shared__ __attribute__((aligned(16))) u32x4 x6w_lds[];      // [3 stages][4 sets][4 tiles][3 pieces][64 lanes]
    const int ntn = N >> 7, ntiles = ntn * (M >> 7);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nblk = K >> 5;
    // A workgroup walks tiles blockIdx.x, + gridDim.x, ... (the launch takes one workgroup per CU when there are more tiles than CUs): both roles run
    // 1 + nblk barriers per tile, and the producers' last barrier of a tile is the one behind the consumers' last read of it - every stage is free when
    // they start on the next tile, whose first blocks are requested and cut while the consumers still store the finished tile (the prologue and the
    // epilogue of a tile were ~8 us of the 22 - 28 us a 11 / 16-block tile takes).  gridDim.x is a multiple of 8 then: a workgroup's tiles stay on its XCD.
    if (wave >= 4) {
        const int ps = wave & 3;
#pragma unroll 1
        for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
            const int tile = fn_xcd_remap(v, ntiles);                    // consecutive tiles (same rows of A) on one XCD
            const int mb = (tile / ntn) * 128, nb = (tile % ntn) * 128;
            if (ps < 2) x6w_produce_nt<false>(x6w_lds, A + (long)(mb + 64 * ps) * lda, lda, ps, lane, nblk);
            else x6w_produce_nt<true>(x6w_lds, B + (long)(nb + 64 * (ps - 2)) * ldb, ldb, ps, lane, nblk);
        }
        return;
    }
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
#pragma unroll 1
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        const int tile = fn_xcd_remap(v, ntiles);
        const int mb = (tile / ntn) * 128, nb = (tile % ntn) * 128;
        f32x4 acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
        x6w_consume(x6w_lds, wm, wn, lane, nblk, acc);
        // D[(l >> 4) * 4 + rr][l & 15] of tile (a, b) = row mb + 64 wm + 4 ((l >> 4) * 4 + rr) + a, column nb + 64 wn + 4 (l & 15) + b
        const int col = nb + 64 * wn + 4 * li;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + col);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int row = mb + 64 * wm + 4 * (lg * 4 + rr) + a;
                float* cp = C + (long)row * ldc + col;
                f32x4 o;
#pragma unroll
                for (int b = 0; b < 4; ++b) o[b] = alpha * acc[a][b][rr] + bv[b];
                if (beta != 0.f) {
                    const f32x4 old = *reinterpret_cast<const f32x4*>(cp);
#pragma unroll
                    for (int b = 0; b < 4; ++b) o[b] += beta * old[b];
                }
                *reinterpret_cast<f32x4*>(cp) = o;
            }
    }
}

// ---------------------------------------------------------------------------------------------------------
// The producer / consumer bf16 x 6 product on 128 x 256 output tiles (round 6).  Measured on gemm_tn_x6w_kernel (profiles/r06_gemm_tn_x6_*.txt): its
// consumers alone run at 0.88 of the MFMA peak and the producers' VALU work costs next to nothing beside them, but the producers' LOADS do not
// arrive faster than ~14.5 bytes per clock and CU (25 M 128-byte requests per dW_hh product at ~400 cycles each against what a CU's L1 keeps in
// flight: 419 us with the MFMAs switched off, whatever the prefetch depth) - a 128 x 128 tile needs 32 KB per 1536 MFMA cycles = 21 bytes per clock.
// A 128 x 256 tile needs (128 + 256) x 32 x 4 = 48 KB per 3072 MFMA cycles = 16 bytes per clock.
//   workgroup = 8 waves; LDS = [2 stages][6 sets: A columns mb + 64 s (s = 0, 1), B columns nb + 64 (s - 2) (s = 2..5)][4 tiles][3 pieces][64 lanes]
//   [8 bf16] = 144 KB.  Producer wave p cuts set p (8 loads of 16 bytes per lane and block) and one k half of set 4 + (p >> 1) (4 loads: lane (i, lg)
//   takes rows 8 (2 (p & 1) + (lg >> 1)) + 4 (lg & 1) + j of its column quad, two lanes fill one 16-byte operand slot with ds_write_b64).
//   Consumer wave (wm, wn) = 64 rows x 128 columns: 4 x 8 accumulator tiles (128 registers), the A triples of its 4 row tiles resident (48
//   registers), the B triples of one column tile per buffer, 4 buffers.  Column tiles 0..5 run "c-major" (24 MFMAs each, B[b + 2] read meanwhile);
//   then every read of the stage has landed -> BARRIER (the producers arrive with the next block cut) -> column tiles 6, 7 run "a-major" and behind
//   the last MFMA of a row tile its A triples of the NEXT block are read into the same registers, B[0], B[1] of the next block beside them: no LDS
//   latency is exposed, one barrier per block, and neither side waits at it unless the other one is late.
// Same products in the same order per accumulator as gemm_tn_x6_kernel: bit-identical on K ranges of whole 32-k blocks.
// ---------------------------------------------------------------------------------------------------------
constexpr int X6V_STAGE = 6 * X6W_SET;           // u32x4 vectors per stage (72 KB)
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
// element e of four float4 vectors (4 consecutive k of one column) -> the three pieces as 4 bf16 each
template <bool RN>
FN_DEVINL void fn_split4(const f32x4 (&v)[4], int e, u32x2& h, u32x2& m, u32x2& l) {
    float x[4], hi[4], r1[4], mi[4], r2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = v[j][e];
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = RN ? fn_rn16(x[j]) : fn_top16(x[j]); r1[j] = x[j] - hi[j]; }
#pragma unroll
    for (int j = 0; j < 4; ++j) { mi[j] = RN ? fn_rn16(r1[j]) : fn_top16(r1[j]); r2[j] = r1[j] - mi[j]; }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        h[j] = fn_pack_top16(hi[2 * j], hi[2 * j + 1]);
        m[j] = fn_pack_top16(mi[2 * j], mi[2 * j + 1]);
        l[j] = fn_pack_top16(r2[2 * j], r2[2 * j + 1]);
    }
}

#ifndef X6V_NS
#define X6V_NS 2
#endif
// producer wave: full set `fs` from (Pf, ldf, colf) [RNF = rounded pieces], k half `kh` of set `hs` from (Ph, ldh, colh) [always B: rounded].
// Plain loads counted by the compiler, steady-state trips straight-line (see x6w_produce): 12 loads per block, `s_waitcnt vmcnt(12 (NS - 1))`.
template <bool RNF>
FN_DEVINL void x6v_produce(u32x4* __restrict__ lds, const float* __restrict__ Pf, long ldf, long colf, int fs, const float* __restrict__ Ph, long ldh,
                           long colh, int hs, int kh, int lane, int kbeg, int kend, int nblk) {
    constexpr int NS = X6V_NS;
    const int lg = lane >> 4, li = lane & 15;
    f32x4 fa[NS][8], fh[NS][4];
    const int hrow = 8 * (2 * kh + (lg >> 1)) + 4 * (lg & 1);      // first row (inside a block) of this lane's four half-set rows
    auto gload = [&](auto SET, int blk) __attribute__((always_inline)) {              // a whole block inside the matrix
#ifndef X6W_EXP_NOLOAD
        constexpr int set = decltype(SET)::value;
        const long kb = (long)kbeg + 32 * blk;
        const float* p0 = Pf + (kb + 8 * lg) * ldf + colf;
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[set][j] = x6w_ld(p0 + j * ldf);
        const float* p1 = Ph + (kb + hrow) * ldh + colh;
#pragma unroll
        for (int j = 0; j < 4; ++j) fh[set][j] = x6w_ld(p1 + j * ldh);
#endif
    };
    auto gload_safe = [&](auto SET, int blk) __attribute__((always_inline)) {         // any block: rows beyond the matrix are read from its last row
#ifndef X6W_EXP_NOLOAD
        constexpr int set = decltype(SET)::value;
        const long k0 = (long)kbeg + 32 * blk + 8 * lg, k1 = (long)kbeg + 32 * blk + hrow;
#pragma unroll
        for (int j = 0; j < 8; ++j) fa[set][j] = x6w_ld(Pf + min(k0 + j, (long)kend - 1) * ldf + colf);
#pragma unroll
        for (int j = 0; j < 4; ++j) fh[set][j] = x6w_ld(Ph + min(k1 + j, (long)kend - 1) * ldh + colh);
#endif
    };
    auto cut = [&](auto SET, int blk, int stage, bool zero_tail) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        if (zero_tail) {                                 // rows beyond the matrix count as zeros (selects, no branch)
            const long k0 = (long)kbeg + 32 * blk + 8 * lg, k1 = (long)kbeg + 32 * blk + hrow;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = k0 + j < kend;
#pragma unroll
                for (int e = 0; e < 4; ++e) fa[set][j][e] = in ? fa[set][j][e] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = k1 + j < kend;
#pragma unroll
                for (int e = 0; e < 4; ++e) fh[set][j][e] = in ? fh[set][j][e] : 0.f;
            }
        }
#ifdef X6W_EXP_NOCUT
        if (blk > 0) {                                   // experiment: the loads stay (their registers are "used"), no split, no LDS write
#pragma unroll
            for (int j = 0; j < 8; ++j) fn_keep(fa[set][j]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fn_keep(fh[set][j]);
            return;
        }
#endif
        u32x4* dst = lds + stage * X6V_STAGE + fs * X6W_SET + lane;
#ifdef X6W_EXP_NOVALU
        if (blk > 0) {                                   // experiment: the same LDS writes without the split arithmetic
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                dst[(a * 3 + 0) * 64] = __builtin_bit_cast(u32x4, fa[set][2 * a]);
                dst[(a * 3 + 1) * 64] = __builtin_bit_cast(u32x4, fa[set][2 * a + 1]);
                dst[(a * 3 + 2) * 64] = __builtin_bit_cast(u32x4, fa[set][(2 * a + 2) & 7]);
            }
            u32x2* dh0 = reinterpret_cast<u32x2*>(lds + stage * X6V_STAGE + hs * X6W_SET + li + 16 * (2 * kh + (lg >> 1))) + (lg & 1);
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const u32x4 w = __builtin_bit_cast(u32x4, fh[set][a]);
                dh0[((a * 3 + 0) * 64) * 2] = (u32x2){w[0], w[1]};
                dh0[((a * 3 + 1) * 64) * 2] = (u32x2){w[2], w[3]};
                dh0[((a * 3 + 2) * 64) * 2] = (u32x2){w[1], w[2]};
            }
            return;
        }
#endif
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            bf16x8 h, m, l;
            fn_split8<RNF>(fa[set], a, h, m, l);
#ifdef X6W_EXP_NODSW
            if (blk > 0) { asm volatile("" ::"v"(h), "v"(m), "v"(l)); continue; }
#endif
            dst[(a * 3 + 0) * 64] = __builtin_bit_cast(u32x4, h);
            dst[(a * 3 + 1) * 64] = __builtin_bit_cast(u32x4, m);
            dst[(a * 3 + 2) * 64] = __builtin_bit_cast(u32x4, l);
        }
        // half set: operand slot of MFMA lane (i = li, g = 2 kh + (lg >> 1)), its low or high 8 bytes (k 8 g + 4 (lg & 1) ..)
        u32x2* dh = reinterpret_cast<u32x2*>(lds + stage * X6V_STAGE + hs * X6W_SET + li + 16 * (2 * kh + (lg >> 1))) + (lg & 1);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u32x2 h, m, l;
            fn_split4<true>(fh[set], a, h, m, l);
#ifdef X6W_EXP_NODSW
            if (blk > 0) { asm volatile("" ::"v"(h), "v"(m), "v"(l)); continue; }
#endif
            dh[((a * 3 + 0) * 64) * 2] = h;
            dh[((a * 3 + 1) * 64) * 2] = m;
            dh[((a * 3 + 2) * 64) * 2] = l;
        }
    };
    // prologue: block 0 cut into stage 0, blocks 1 .. NS - 1 requested
    x6w_for<NS>([&](auto I) __attribute__((always_inline)) {
        if (decltype(I)::value < nblk) gload_safe(I, decltype(I)::value);
    });
    cut(x6w_ic<0>{}, 0, 0, true);
    x6w_barrier_p();
    int t = 0;
    // steady state: trip t (t % NS == r) requests block t + NS -> set r and cuts block t + 1 (set (r + 1) % NS, stage (t + 1) & 1); whole blocks only
    // request and cut interleaved (see x6w_produce): two loads in front of every quarter of the full set's cut, one in front of every quarter of the half set's
    auto fused = [&](auto LS, int lblk, auto CS, int stage) __attribute__((always_inline)) {
        constexpr int ls = decltype(LS)::value, cs = decltype(CS)::value;
        const long kb = (long)kbeg + 32 * lblk;
        const float* p0 = Pf + (kb + 8 * lg) * ldf + colf;
        const float* p1 = Ph + (kb + hrow) * ldh + colh;
        u32x4* dst = lds + stage * X6V_STAGE + fs * X6W_SET + lane;
        u32x2* dh = reinterpret_cast<u32x2*>(lds + stage * X6V_STAGE + hs * X6W_SET + li + 16 * (2 * kh + (lg >> 1))) + (lg & 1);
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef X6W_EXP_NOLOAD
            fa[ls][2 * a] = x6w_ld(p0 + (2 * a) * ldf);
            fa[ls][2 * a + 1] = x6w_ld(p0 + (2 * a + 1) * ldf);
#endif
            bf16x8 h, m, l;
            fn_split8<RNF>(fa[cs], a, h, m, l);
            dst[(a * 3 + 0) * 64] = __builtin_bit_cast(u32x4, h);
            dst[(a * 3 + 1) * 64] = __builtin_bit_cast(u32x4, m);
            dst[(a * 3 + 2) * 64] = __builtin_bit_cast(u32x4, l);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#ifndef X6W_EXP_NOLOAD
            fh[ls][a] = x6w_ld(p1 + a * ldh);
#endif
            u32x2 h, m, l;
            fn_split4<true>(fh[cs], a, h, m, l);
            dh[((a * 3 + 0) * 64) * 2] = h;
            dh[((a * 3 + 1) * 64) * 2] = m;
            dh[((a * 3 + 2) * 64) * 2] = l;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
#pragma unroll 1
    for (; t + 2 * NS < nblk; t += NS) {
        x6w_for<NS>([&](auto R) __attribute__((always_inline)) {
            constexpr int r = decltype(R)::value;
            if (X6W_INTERLEAVE) {
                fused(R, t + r + NS, x6w_ic<(r + 1) % NS>{}, (t + r + 1) & 1);
            } else {
                gload(R, t + r + NS);
                cut(x6w_ic<(r + 1) % NS>{}, t + r + 1, (t + r + 1) & 1, false);
            }
            x6w_barrier_p();
        });
    }
    // the last <= 2 NS trips (t is a multiple of NS here)
    x6w_for<2 * NS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, r = i % NS;
        if (t + i < nblk) {
            if (t + i + NS < nblk) gload_safe(x6w_ic<r>{}, t + i + NS);
            if (t + i + 1 < nblk) cut(x6w_ic<(r + 1) % NS>{}, t + i + 1, (t + i + 1) & 1, true);
            x6w_barrier_p();
        }
    });
}

// one (output tile, K range) of gemm_tn_x6v_kernel: both roles run 1 + nblk barriers, and behind the last one no stage is read any more (the reads in flight fetch
// values nobody uses): the next item of a workgroup that walks several of them may be cut into the stages at once
FN_DEVINL void gemm_tn_x6v_kernel_item(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                              const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                              const float* __restrict__ A2, long lda2, int msplit, u32x4* __restrict__ x6v_lds, int ntn, int tile, int zk) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int mb = (tile / ntn) * 128, nb = (tile % ntn) * 256;          // the workgroup's output tile
    const int li = lane & 15, lg = lane >> 4;
    const int kbeg = zk * ksplit_len, kend = min(K, kbeg + ksplit_len);
    const int nblk = (kend - kbeg + 31) >> 5;
    if (nblk <= 0) return;                               // (the host never launches an empty K range)

    if (wave >= 4) {
        // ---- producer p: set p (p < 2: A columns mb + 64 p; else B columns nb + 64 (p - 2)) and k half p & 1 of set 4 + (p >> 1) (B columns nb + 128 + 64 (p >> 1)) ----
        const int p = wave & 3;
        const bool pa = p < 2;
        const float* P = pa ? A : B;
        long pld = pa ? lda : ldb;
        // column offsets of this lane's 16-byte loads, kept inside the columns the operand REALLY has (see gemm_tn_body)
        const int pc0 = pa ? mb + 64 * p : nb + 64 * (p - 2);
        long ncols = pa ? (long)M : (long)N, rel = pc0;
        if (pa && A2 != nullptr) {
            if (pc0 >= msplit) { P = A2; pld = lda2; ncols = (long)M - msplit; rel = pc0 - msplit; }
            else ncols = msplit;
        }
        long pcol = min(rel + 4 * li, ((ncols - 1) >> 2) << 2);
        if (rel >= ncols) pcol = ((ncols - 1) >> 2) << 2;                // a set entirely beyond the matrix: any legal column (never stored)
        const long hrel = (long)nb + 128 + 64 * (p >> 1);
        long hcol = min(hrel + 4 * li, (((long)N - 1) >> 2) << 2);
        if (hrel >= N) hcol = (((long)N - 1) >> 2) << 2;
        if (pa) x6v_produce<false>(x6v_lds, P, pld, pcol, p, B, ldb, hcol, 4 + (p >> 1), p & 1, lane, kbeg, kend, nblk);
        else x6v_produce<true>(x6v_lds, P, pld, pcol, p, B, ldb, hcol, 4 + (p >> 1), p & 1, lane, kbeg, kend, nblk);
        return;
    }

    // ---- consumer (wm, wn): rows mb + 64 wm (set wm), columns nb + 128 wn (sets 2 + 2 wn, 3 + 2 wn) ----
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 Af[4][3], Bf[4][3];
    const u32x4* lA = x6v_lds + wm * X6W_SET + lane;
    const u32x4* lB = x6v_lds + (2 + 2 * wn) * X6W_SET + lane;        // column tile b = tile b & 3 of set 2 + 2 wn + (b >> 2): (b * 3 + piece) * 64 vectors on
    auto rdA = [&](int stage, int a) __attribute__((always_inline)) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Af[a][pc] = __builtin_bit_cast(bf16x8, lA[stage * X6V_STAGE + (a * 3 + pc) * 64]);
    };
    auto rdB = [&](int stage, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Bf[b & 3][pc] = __builtin_bit_cast(bf16x8, lB[stage * X6V_STAGE + (b * 3 + pc) * 64]);
    };
    // the six products of a 32-k block, smallest first (piece 0 = hi, 1 = mid, 2 = lo): lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    x6w_barrier_p();                                     // stage 0 holds block 0
    rdB(0, 0);
#pragma unroll
    for (int a = 0; a < 4; ++a) rdA(0, a);
    rdB(0, 1);
#pragma unroll 1
    for (int t = 0; t < nblk; ++t) {
        const int st = t & 1, sn = st ^ 1;
        // column tiles 0..5, c-major: 24 MFMAs each, the B triples of column tile b + 2 read meanwhile (sched_barrier: nothing migrates between the
        // phases - left alone the compiler collects the reads of several phases in one place and waits for all of them)
#pragma unroll
        for (int b = 0; b < 6; ++b) {
            rdB(st, b + 2);
#ifndef X6W_EXP_NOMMA
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[a][PA[c]], Bf[b & 3][PB[c]], acc[a][b], 0, 0, 0);
#endif
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // every read of this stage has landed; the producers arrive with the next block cut into the other stage
        x6w_barrier_p();
        // column tiles 6, 7, a-major; behind a row tile's last MFMA its A triples of the NEXT block are read, B[0], B[1] of the next block beside them
        // (behind the last block: reads of a stage nobody uses)
        auto joint = [&](auto AA) __attribute__((always_inline)) {
            constexpr int a = decltype(AA)::value;
#ifndef X6W_EXP_NOMMA
#pragma unroll
            for (int c = 0; c < 6; ++c)
#pragma unroll
                for (int b = 6; b < 8; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(Af[a][PA[c]], Bf[b & 3][PB[c]], acc[a][b], 0, 0, 0);
#endif
        };
        rdB(sn, 0);
        joint(x6w_ic<0>{});
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        rdA(sn, 0);
        joint(x6w_ic<1>{});
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        rdA(sn, 1);
        rdB(sn, 1);
        joint(x6w_ic<2>{});
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        rdA(sn, 2);
        joint(x6w_ic<3>{});
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        __builtin_amdgcn_sched_barrier(0);
        rdA(sn, 3);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int m0 = mb + 64 * wm, n0 = nb + 128 * wn;
    if (slabs != nullptr && (N & 3) == 0) {              // split-K slabs: a lane's four column tiles of a set are four consecutive columns - 16-byte stores
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + 4 * (lg * 4 + r) + a;
                if (row >= M) continue;
#pragma unroll
                for (int bs = 0; bs < 2; ++bs) {
                    const int col = n0 + 64 * bs + 4 * li;
                    if (col >= N) continue;
                    const f32x4 v = {acc[a][4 * bs][r], acc[a][4 * bs + 1][r], acc[a][4 * bs + 2][r], acc[a][4 * bs + 3][r]};
                    *reinterpret_cast<f32x4*>(slabs + ((long)zk * M + row) * N + col) = v;
                }
            }
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 4 * (lg * 4 + r) + a;
            if (row >= M) continue;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int col = n0 + 64 * (b >> 2) + 4 * li + (b & 3);
                if (col >= N) continue;
                const float v = acc[a][b][r];
                if (slabs) {
                    slabs[((long)zk * M + row) * N + col] = v;
                } else {
                    float o = alpha * v;
                    if (bias) o += bias[col];
                    if (beta != 0.f) o += beta * C[(long)row * ldc + col];
                    C[(long)row * ldc + col] = o;
                }
            }
        }
}

__global__ __launch_bounds__(X6W_NT) void gemm_tn_x6v_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                              const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                              const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                              const float* __restrict__ A2, long lda2, int msplit) {
    extern __shared__ __attribute__((aligned(16))) u32x4 x6v_lds[];      // [2 stages][6 sets][4 tiles][3 pieces][64 lanes]
    const int ntn = (N + 255) / 256, ntm = (M + 127) / 128;
    if (gridDim.z == 1 && slabs != nullptr) {            // K ranges dealt to the XCDs (see gemm_tn_body); a workgroup walks items blockIdx.x, + gridDim.x, ...
        const int S = (K + ksplit_len - 1) / ksplit_len, items = S * ntn * ntm;      // (gridDim.x < items only when it is a multiple of 8: the items of a workgroup stay on its XCD)
#pragma unroll 1
        for (int v = blockIdx.x; v < items; v += gridDim.x) {
            const int c = v & 7, q = v >> 3;
            gemm_tn_x6v_kernel_item(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit, x6v_lds, ntn, q % (ntn * ntm), c * (S >> 3) + q / (ntn * ntm));
        }
    } else {
        gemm_tn_x6v_kernel_item(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit, x6v_lds, ntn, fn_xcd_remap(blockIdx.x, ntn * ntm), blockIdx.z);
    }
}

// ---------------------------------------------------------------------------------------------------------
// "NT" GEMM without LDS for the short-K products beside the decoder scans (layer-2 input projection gx2 = hx0 W_ih2^T and the input
// gradient dhx0 = dgx2 W_ih2 through the transposed weight image): C[M][N] = alpha * sum_k A[m][k] B[n][k] (+ bias, + beta C), both
// operands K-contiguous.  The LDS-staged kernel spends its 16 K tiles of a K = 512 product on prologue / barrier / epilogue (MFMA pipe
// 54 % busy, 92 TFLOP/s); here a lane reads its MFMA operands straight from memory: lane (i = l & 15, g = l >> 4) loads the float4
// A[row i of a 16-row tile][k0 + 4g .. 4g + 3]; MFMA j (j = 0..3) of the step takes element j of every lane, i.e. k slot g of MFMA j is
// k0 + 4g + j - a permutation of the 16 k values of the step, the same for A and B, so the sum is the same set of products in a fixed
// order.  One wave = 64 x 64 outputs (4 x 4 tiles), B tiles interleaved (tile b = columns n0 + 4i + b) so that a lane's four b values
// are one float4 store; PF steps of 16 k (8 loads each) in flight, counted vmcnt; loads use a scalar base that advances 64 bytes per
// step + a fixed 32-bit lane offset (no vector address arithmetic in the loop).  Requires M, N multiples of 128, K a multiple of 16,
// 16-byte aligned operands with ld % 4 == 0 and < 4 GB spans (fn_gemm_f32 falls back to the staged kernel otherwise).
// ---------------------------------------------------------------------------------------------------------
FN_DEVINL void fn_gld4_s(f32x4& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

template <int PF, int WGS>
__global__ __launch_bounds__(NT, WGS) void gemm_nt_direct_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                            const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                            const float* __restrict__ bias) {
    const int ntn = N >> 7, ntm = M >> 7;
    const int tile = fn_xcd_remap(blockIdx.x, ntn * ntm);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (tile / ntn) * 128 + (wave >> 1) * 64, n0 = (tile % ntn) * 128 + (wave & 1) * 64;
    const int li = lane & 15, lg = lane >> 4;
    unsigned oa[4], ob[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        oa[a] = (unsigned)(((long)(m0 + 16 * a + li) * lda + 4 * lg) * 4);
        ob[a] = (unsigned)(((long)(n0 + 4 * li + a) * ldb + 4 * lg) * 4);
    }
    f32x4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nks = K >> 4;                          // steps of 16 k
    const int nmain = nks / PF * PF;
    const float* pa = A;
    const float* pb = B;
    f32x4 fa[PF][4], fb[PF][4];
    auto load = [&](int set) {
#pragma unroll
        for (int a = 0; a < 4; ++a) fn_gld4_s(fa[set][a], oa[a], pa);
#pragma unroll
        for (int b = 0; b < 4; ++b) fn_gld4_s(fb[set][b], ob[b], pb);
        pa += 16;
        pb += 16;
    };
    auto mma = [&](int u) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(fa[u][a], j), f4c(fb[u][b], j), acc[a][b], 0, 0, 0);
    };
    if (nmain > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s) load(s);
        for (int base = 0; base + PF < nmain; base += PF) {   // steady state: every step refills its own ring slot
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                fn_wait_vm<8 * (PF - 1)>();
                mma(u);
                load(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {                        // last PF steps: everything has been requested
            if (u == 0) fn_wait_vm<8 * (PF - 1)>();
            else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? 8 * (PF - 2) : 0)>();
            else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? 8 * (PF - 3) : 0)>();
            else fn_wait_vm<0>();
            mma(u);
        }
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int a = 0; a < 4; ++a) { fn_keep(fa[s][a]); fn_keep(fb[s][a]); }
    }
    for (int ks = nmain; ks < nks; ++ks) {                    // < PF leftover steps, unpipelined
        load(0);
        fn_wait_vm<0>();
        mma(0);
#pragma unroll
        for (int a = 0; a < 4; ++a) { fn_keep(fa[0][a]); fn_keep(fb[0][a]); }
    }
    f32x4 bv = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * li);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long row = m0 + 16 * a + 4 * lg + r;
            float* cp = C + row * ldc + n0 + 4 * li;
            f32x4 o = (f32x4){acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]} * alpha + bv;
            if (beta != 0.f) o += beta * *reinterpret_cast<const f32x4*>(cp);
            *reinterpret_cast<f32x4*>(cp) = o;
        }
}

__global__ __launch_bounds__(NT) void gemm_tn_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                                                     const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                                                     const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                                                     const float* __restrict__ A2, long lda2, int msplit) {
    gemm_tn_body<8>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit);
}

// The same product for the idle MFMA cycles of a weight-stationary scan: <= 128 vector registers per wavefront (4 wavefronts per SIMD
// requested), so that one fits beside a scan wavefront (336-376 of the SIMD's 512 registers); 4 k-steps of operands in flight instead
// of 8 - beside a scan it is issue-starved, not latency-bound.  s_setprio 0 (the scans run at 3): it only takes what they leave.
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gemm_tn_lean_kernel(int M, int N, int K, float alpha, const float* __restrict__ A, long lda,
                         const float* __restrict__ B, long ldb, float beta, float* __restrict__ C, long ldc,
                         const float* __restrict__ bias, int ksplit_len, float* __restrict__ slabs,
                         const float* __restrict__ A2, long lda2, int msplit) {
    gemm_tn_body<4>(M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, ksplit_len, slabs, A2, lda2, msplit);
}

// ---------------------------------------------------------------------------------------------------------
// Output head of the teacher-forced decoder in ONE kernel: logits = h W^T + b (512 -> 342), log-softmax over the vocabulary, NLL of the
// target token and the gradient seed grad_scale * (softmax - onehot) - gmm_model.py:137 + trainer_gmm.py:131-132 and their autograd.
// The logits never reach HBM (unfused: 90 MB written by the GEMM, read and rewritten by the softmax kernel at the benchmark shape).
// Workgroup = 64 rows x all vocabulary columns (384 = 24 column tiles of 16, columns >= V are padding), waves 2 x 2: wave (wm, wn) owns
// rows [32 wm, 32 wm + 32) x columns [192 wn, 192 wn + 192) - 14 LDS fragment reads per 48 MFMAs (a 4 x 1 layout with complete rows per
// wave needs 25 and was LDS-bound: 60 TFLOP/s); the row maximum / sum of the two column halves meet through LDS.  Same k order as
// gemm_kernel in the staged path; the LDS-free path (aligned operands, K % 16 == 0) permutes k inside a 16-k step.  The gradient rows leave through LDS as aligned float4 rows.
constexpr int OH_BM = 64, OH_BN = 384, OH_BK = 16, OH_LDT = OH_BN + 4;
// wave layout of a 64-row x 384-column workgroup: OH_WN column groups x (4 / OH_WN) row groups.  1 x 4 (round 6): a wave = 64 rows x 96 columns loads 4 + 6
// operand tiles per 96 MFMAs, 2 x 2 (up to round 5: 32 rows x 192 columns) 2 + 12 - the LDS-free loop is bound by what a CU's L1 keeps in flight
#ifndef OH_WN
#define OH_WN 4
#endif
constexpr int OH_WM = 4 / OH_WN;
size_t out_head_lds_bytes() {
    const size_t stage = (size_t)2 * (Stage<OH_BM, OH_BK, true, NT>::WORDS + Stage<OH_BN, OH_BK, true, NT>::WORDS) * sizeof(float);
    const size_t rows = (size_t)(OH_BM / 2) * OH_LDT * sizeof(float) + 2 * OH_WN * OH_BM * sizeof(float);      // half the rows at a time + row statistics
    return stage > rows ? stage : rows;
}

__global__ __launch_bounds__(NT, 2) void out_head_kernel(const float* __restrict__ h, long ldh, const float* __restrict__ W, long ldw,
                                                         const float* __restrict__ bias, int R, int V, int K, int B, int T,
                                                         const int* __restrict__ target, float grad_scale, float* __restrict__ nll_rows,
                                                         float* __restrict__ dlogits, long ld) {
    constexpr int TM = OH_BM / OH_WM / 16, TN = OH_BN / OH_WN / 16;
    using SA = Stage<OH_BM, OH_BK, true, NT>;
    using SB = Stage<OH_BN, OH_BK, true, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int m0 = blockIdx.x * OH_BM;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / OH_WN, wn = wave % OH_WN;
    const RowsPlain ra{m0, R}, rb{0, V};
    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int nk = (K + OH_BK - 1) / OH_BK;
    if (fn_aligned16(h, ldh) && fn_aligned16(W, ldw) && (K % 16) == 0 && (long)R * ldh < (1L << 30) && (long)V * ldw < (1L << 30)) {
        // LDS-free K loop (as gemm_nt_direct_kernel): lane (i, g) loads the float4 [row i][k0 + 4g ..] of its 2 row tiles and 12 weight-row
        // tiles; MFMA j of a 16-k step takes element j of every lane.  14 loads per 96 MFMAs, two steps in flight, no barrier in the loop
        // (MFMA pipe 54 % busy with the LDS-staged loop: 32 K tiles of staging and barriers per workgroup)
        const int li = lane & 15, lg = lane >> 4;
        unsigned oa[TM], ob[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) oa[m] = (unsigned)(((long)min(m0 + wm * (TM * 16) + 16 * m + li, R - 1) * ldh + 4 * lg) * 4);
#pragma unroll
        for (int n = 0; n < TN; ++n) ob[n] = (unsigned)(((long)min(wn * (TN * 16) + 16 * n + li, V - 1) * ldw + 4 * lg) * 4);
        constexpr int PFD = 2, NL = TM + TN;
        f32x4 fa[PFD][TM], fb[PFD][TN];
        const float* pa = h;
        const float* pb = W;
        auto load = [&](int set) {
#pragma unroll
            for (int m = 0; m < TM; ++m) fn_gld4_s(fa[set][m], oa[m], pa);
#pragma unroll
            for (int n = 0; n < TN; ++n) fn_gld4_s(fb[set][n], ob[n], pb);
            pa += 16;
            pb += 16;
        };
        auto mma = [&](int u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4c(fa[u][m], j), f4c(fb[u][n], j), acc[m][n], 0, 0, 0);
        };
        const int nks = K >> 4, nmain = nks / PFD * PFD;
        if (nmain > 0) {
            load(0);
            load(1);
            for (int base = 0; base + PFD < nmain; base += PFD) {
#pragma unroll
                for (int u = 0; u < PFD; ++u) {
                    fn_wait_vm<NL*(PFD - 1)>();
                    mma(u);
                    load(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fn_wait_vm<NL>();
            mma(0);
            fn_wait_vm<0>();
            mma(1);
        }
        if (nmain < nks) {
            load(0);
            fn_wait_vm<0>();
            mma(0);
        }
#pragma unroll
        for (int s_ = 0; s_ < PFD; ++s_) {
#pragma unroll
            for (int m = 0; m < TM; ++m) fn_keep(fa[s_][m]);
#pragma unroll
            for (int n = 0; n < TN; ++n) fn_keep(fb[s_][n]);
        }
    } else if (SA::can_fast(h, ldh, ra, K) && SB::can_fast(W, ldw, rb, K)) {
        auto loadA = [&](int k0, SA& st) { st.load_fast(h, ldh, ra, k0); };
        auto loadB = [&](int k0, SB& st) { st.load_fast(W, ldw, rb, k0); };
        fn_kloop<PF_DEPTH, TM, TN, OH_BK, SA, SB>(smem, nk, loadA, loadB, wm * (TM * 16), wn * (TN * 16), lane, acc);
    } else {
        auto loadA = [&](int k0, SA& st) { st.load_checked(h, ldh, ra, k0, K); };
        auto loadB = [&](int k0, SB& st) { st.load_checked(W, ldw, rb, k0, K); };
        fn_kloop<PF_DEPTH, TM, TN, OH_BK, SA, SB>(smem, nk, loadA, loadB, wm * (TM * 16), wn * (TN * 16), lane, acc);
    }
    // (the K loop ended with a barrier: the staging buffers are free)
    // D[row = 16 m + (lane>>4)*4 + i][col = 16 TN wn + 16 n + (lane&15)]; row statistics of the OH_WN column groups meet in stat[]
    float* stat = smem + (OH_BM / 2) * OH_LDT;          // [OH_WN groups][64 rows] maxima, then [OH_WN][64] sums
    const int cj = lane & 15, rq = (lane >> 4) * 4, c0 = wn * (TN * 16), r0 = wm * (TM * 16);
    float mxr[TM][4], lser[TM][4];
#pragma unroll
    for (int n = 0; n < TN; ++n) {
        const int col = c0 + 16 * n + cj;
        const float bv = col < V ? bias[col] : 0.f;
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] += bv;
    }
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float mx = -INFINITY;
#pragma unroll
            for (int n = 0; n < TN; ++n)
                if (c0 + 16 * n + cj < V) mx = fmaxf(mx, acc[m][n][i]);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
            if (cj == 0) stat[wn * OH_BM + r0 + 16 * m + rq + i] = mx;
        }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = r0 + 16 * m + rq + i;
            float mx = stat[rl];
#pragma unroll
            for (int g = 1; g < OH_WN; ++g) mx = fmaxf(mx, stat[g * OH_BM + rl]);
            mxr[m][i] = mx;
            float sum = 0.f;
#pragma unroll
            for (int n = 0; n < TN; ++n)
                if (c0 + 16 * n + cj < V) sum += expf(acc[m][n][i] - mx);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
            if (cj == 0) stat[(OH_WN + wn) * OH_BM + rl] = sum;
        }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rl = r0 + 16 * m + rq + i;
            const int row = m0 + rl;
            float tot = stat[OH_WN * OH_BM + rl];                                               // group 0 + 1 + ...: the same order in every wave
#pragma unroll
            for (int g = 1; g < OH_WN; ++g) tot += stat[(OH_WN + g) * OH_BM + rl];
            const float lse = mxr[m][i] + logf(tot);
            lser[m][i] = lse;
            const int rc = min(row, R - 1);
            const int tg = target[(long)(rc % B) * T + rc / B];
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = c0 + 16 * n + cj;
                const float l = acc[m][n][i] - lse;
                if (col == tg && row < R && nll_rows) nll_rows[row] = -l;
                acc[m][n][i] = col < V ? grad_scale * (expf(l) - (col == tg ? 1.0f : 0.0f)) : 0.f;
            }
        }
    if (!dlogits) return;
    // gradient rows -> LDS -> aligned float4 rows, 32 rows (one wm) at a time: half the LDS, two workgroups per CU
    const int nq = (int)(ld >> 2);                      // float4 per output row (ld is a multiple of 4, <= OH_BN)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                // statistics read / previous half written out
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int rb0 = r0 + 16 * m;                 // this row tile's first row inside the 64-row panel (a multiple of 16: one half or the other)
            if ((rb0 >> 5) != half) continue;
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) smem[((rb0 & 31) + rq + i) * OH_LDT + c0 + 16 * n + cj] = acc[m][n][i];
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < 32 * nq; idx += NT) {
            const int r = idx / nq, c4 = idx - r * nq;
            const int row = m0 + half * 32 + r;
            if (row < R) *reinterpret_cast<float4*>(dlogits + (long)row * ld + 4 * c4) = *reinterpret_cast<const float4*>(smem + r * OH_LDT + 4 * c4);
        }
    }
}

// C = alpha * sum_s slabs[s] + beta*C + bias
__global__ void slab_reduce_kernel(const float* __restrict__ slabs, int S, int M, int N, float alpha, float beta,
                                   float* __restrict__ C, long ldc, const float* __restrict__ bias) {
    const long total = (long)M * N;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < S; ++k) s += slabs[k * total + i];
        const int row = i / N, col = i % N;
        float o = alpha * s;
        if (bias) o += bias[col];
        if (beta != 0.f) o += beta * C[(long)row * ldc + col];
        C[(long)row * ldc + col] = o;
    }
}

// the same for N % 4 == 0, 16-byte aligned slabs / C / bias: four columns per thread, eight slab loads in flight (same summation order per element: slab 0, 1, ...)
__global__ void slab_reduce4_kernel(const float* __restrict__ slabs, int S, int M, int N, float alpha, float beta,
                                    float* __restrict__ C, long ldc, const float* __restrict__ bias) {
    const long total4 = (long)M * N / 4, total = (long)M * N;
    const int n4 = N / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        const float* p = slabs + 4 * i;
        int k = 0;
        for (; k + 8 <= S; k += 8) {
            f32x4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const f32x4*>(p + (long)(k + u) * total);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < S; ++k) s += *reinterpret_cast<const f32x4*>(p + (long)k * total);
        const int row = (int)(i / n4), col = (int)(i % n4) * 4;
        f32x4 o = alpha * s;
        if (bias) o += *reinterpret_cast<const f32x4*>(bias + col);
        float* cp = C + (long)row * ldc + col;
        if (beta != 0.f) o += beta * *reinterpret_cast<const f32x4*>(cp);
        *reinterpret_cast<f32x4*>(cp) = o;
    }
}

// split-K slabs -> C (the float4 form where everything is 16-byte aligned; same summation order per element)
static void launch_slab_reduce(hipStream_t st, const float* slabs, int S, int M, int N, float alpha, float beta, float* C, long ldc, const float* bias) {
    const bool v4 = (N & 3) == 0 && (ldc & 3) == 0 && (((((uintptr_t)slabs) | ((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0);
    const long total = v4 ? (long)M * N / 4 : (long)M * N;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (v4) hipLaunchKernelGGL(slab_reduce4_kernel, dim3(blocks), dim3(256), 0, st, slabs, S, M, N, alpha, beta, C, ldc, bias);
    else hipLaunchKernelGGL(slab_reduce_kernel, dim3(blocks), dim3(256), 0, st, slabs, S, M, N, alpha, beta, C, ldc, bias);
}

__global__ void transpose_kernel(const float* __restrict__ src, int R, int Cc, long src_ld, float* __restrict__ dst, long dst_ld) {
    __shared__ float t[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        t[j][threadIdx.x] = (r < R && c < Cc) ? src[(long)r * src_ld + c] : 0.f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += 8) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < R && c < Cc) dst[(long)c * dst_ld + r] = t[threadIdx.x][j];
    }
}

// partial[chunk][n] = sum over rows of the chunk
__global__ void colsum_partial_kernel(const float* __restrict__ X, int M, int N, long ld, int rows_per_chunk,
                                      float* __restrict__ partial) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int m0 = blockIdx.y * rows_per_chunk, m1 = min(M, m0 + rows_per_chunk);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;      // 4 independent loads in flight per thread
    int m = m0;
    for (; m + 3 < m1; m += 4) {
        s0 += X[(long)m * ld + n];
        s1 += X[(long)(m + 1) * ld + n];
        s2 += X[(long)(m + 2) * ld + n];
        s3 += X[(long)(m + 3) * ld + n];
    }
    for (; m < m1; ++m) s0 += X[(long)m * ld + n];
    partial[(long)blockIdx.y * N + n] = (s0 + s1) + (s2 + s3);
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int chunks, int N, float beta, float* __restrict__ out) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partial[(long)c * N + n];
    out[n] = (beta != 0.f ? beta * out[n] : 0.f) + s;
}

// all small column sums of a step in one launch: workgroup (x, z) = 64 columns [64x, 64x+64) of job z; thread = (column, row phase 0..3)
struct CsArgs { FnColsumJob job[FN_COLSUM_MAX_JOBS]; };
__global__ __launch_bounds__(256) void colsum_multi_kernel(const CsArgs a) {
    const FnColsumJob& J = a.job[blockIdx.z];
    const int n = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    __shared__ float part[4][64];
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (blockIdx.x * 64 < J.N) {
        const int nc = n < J.N ? n : J.N - 1;
        const float* X = J.X + nc;
        int m = ph;
        for (; m + 12 < J.M; m += 16) {                       // 4 independent loads in flight per thread
            s0 += X[(long)m * J.ld];
            s1 += X[(long)(m + 4) * J.ld];
            s2 += X[(long)(m + 8) * J.ld];
            s3 += X[(long)(m + 12) * J.ld];
        }
        for (; m < J.M; m += 4) s0 += X[(long)m * J.ld];
    }
    part[ph][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ph == 0 && n < J.N) {
        const float t = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        J.out[n] = (J.beta != 0.f ? J.beta * J.out[n] : 0.f) + t;
    }
}

__global__ void axpy_kernel(long n, float alpha, const float* __restrict__ x, float* __restrict__ y) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] += alpha * x[i];
}

__global__ void sum_kernel(const float* __restrict__ x, long n, float scale, float* __restrict__ out) {
    __shared__ float red[1024 / 64];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += blockDim.x) s += x[i];
    s = fn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
        out[0] = t * scale;
    }
}

template <int BM, int BN, int BK, int WM, int WN>
int launch_gemm(int ak, int bk, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb, float beta,
                float* C, int ldc, const float* bias, int splitk, float* ws, hipStream_t st) {
    const int ntm = (M + BM - 1) / BM, ntn = (N + BN - 1) / BN;
    int klen = K;
    if (splitk > 1) {
        klen = ((K + splitk - 1) / splitk + BK - 1) / BK * BK;
        splitk = (K + klen - 1) / klen;
    }
    dim3 grid(ntm * ntn, 1, splitk > 1 ? splitk : 1);
    float* slabs = splitk > 1 ? ws : nullptr;
#define FN_GEMM_LAUNCH(AK, BKK)                                                                                            \
    {                                                                                                                      \
        using SA = Stage<BM, BK, AK, NT>;                                                                                  \
        using SB = Stage<BN, BK, BKK, NT>;                                                                                 \
        const size_t sh = 2 * (SA::WORDS + SB::WORDS) * sizeof(float);                                                     \
        hipLaunchKernelGGL((gemm_kernel<BM, BN, BK, WM, WN, AK, BKK>), grid, dim3(NT), sh, st, M, N, K, alpha, A, (long)lda, \
                           B, (long)ldb, beta, C, (long)ldc, bias, klen, slabs);                                           \
    }
    if (ak && bk) FN_GEMM_LAUNCH(true, true)
    else if (ak && !bk) FN_GEMM_LAUNCH(true, false)
    else if (!ak && !bk) FN_GEMM_LAUNCH(false, false)
    else FN_GEMM_LAUNCH(false, true)
#undef FN_GEMM_LAUNCH
    FN_CHECK_LAUNCH();
    if (splitk > 1) {
        launch_slab_reduce(st, slabs, splitk, M, N, alpha, beta, C, (long)ldc, bias);
        FN_CHECK_LAUNCH();
    }
    return FN_OK;
}

}  // namespace

extern "C" {

int fn_out_head_f32(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int T, int V, int H, const int32_t* target,
                    float grad_scale, float* nll_rows, float* dlogits, int ld, void* stream) {
    if (!h || !W || !bias || !target) return FN_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || H <= 0 || ldh < H || ldw < H) return FN_E_SHAPE;
    if (V > OH_BN) return FN_E_UNSUPPORTED;
    if (dlogits && (ld < V || ld > OH_BN || (ld & 3))) return FN_E_SHAPE;
    if (dlogits && (((uintptr_t)dlogits) & 15)) return FN_E_ALIGN;
    const long R = (long)B * T;
    if (R > 0x7fffffff) return FN_E_SHAPE;
    static std::atomic<bool> attr_set[32];         // write-once per device; setting the attribute twice is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    const size_t lds = out_head_lds_bytes();
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(out_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(out_head_kernel, dim3((unsigned)((R + OH_BM - 1) / OH_BM)), dim3(NT), lds, (hipStream_t)stream, h, (long)ldh, W, (long)ldw,
                       bias, (int)R, V, H, B, T, target, grad_scale, nll_rows, dlogits, (long)ld);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_gemm_multi(int a_kmajor, int b_kmajor, const FnGemmJob* jobs, int n_jobs, void* stream) {
    if (!jobs) return FN_E_NULL;
    if (n_jobs <= 0 || n_jobs > GM_MAX_JOBS) return FN_E_COUNT;
    GmArgs a;
    int maxtiles = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const FnGemmJob& d = jobs[j];
        if (!d.C || d.n_seg <= 0 || d.n_seg > GM_MAX_SEG || d.M <= 0 || d.N <= 0 || d.ldc < d.N) return d.C ? FN_E_SHAPE : FN_E_NULL;
        GmJob& J = a.job[j];
        for (int i = 0; i < d.n_seg; ++i) {
            const FnGemmSeg& sg = d.seg[i];
            if (!sg.A || !sg.B) return FN_E_NULL;
            if (sg.K <= 0 || sg.lda < (a_kmajor ? sg.K : d.M) || sg.ldb < (b_kmajor ? sg.K : d.N)) return FN_E_SHAPE;
            J.seg[i] = GmSeg{sg.A, sg.B, sg.lda, sg.ldb, sg.K};
        }
        J.C = d.C; J.bias = d.bias; J.M = d.M; J.N = d.N; J.ldc = d.ldc; J.nseg = d.n_seg; J.beta = d.beta;
        const int tiles = ((d.M + 63) / 64) * ((d.N + 63) / 64);
        maxtiles = tiles > maxtiles ? tiles : maxtiles;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(maxtiles, 1, n_jobs);
#define FN_GM_LAUNCH(AK, BKK)                                                                        \
    {                                                                                                \
        const size_t sh = 2 * (Stage<64, 16, AK, NT>::WORDS + Stage<64, 16, BKK, NT>::WORDS) * sizeof(float); \
        hipLaunchKernelGGL((gemm_multi_kernel<AK, BKK>), grid, dim3(NT), sh, st, a);                 \
    }
    if (a_kmajor && b_kmajor) FN_GM_LAUNCH(true, true)
    else if (a_kmajor && !b_kmajor) FN_GM_LAUNCH(true, false)
    else if (!a_kmajor && !b_kmajor) FN_GM_LAUNCH(false, false)
    else FN_GM_LAUNCH(false, true)
#undef FN_GM_LAUNCH
    FN_CHECK_LAUNCH();
    return FN_OK;
}

// grid of a TN launch: 1-D (K ranges dealt to the XCDs, see gemm_tn_body) when the K ranges divide by 8, else tiles x 1 x K ranges
// compute units of the current device (write-once per device, as gru_persist.hip's cu_count)
static int gemm_cu_count() {
    static std::atomic<int> n[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return 0;
    int c = n[dev].load(std::memory_order_acquire);
    if (c == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        c = prop.multiProcessorCount;
        n[dev].store(c, std::memory_order_release);
    }
    return c;
}

static dim3 tn_grid(int tiles, int splitk) {
    return splitk > 1 && (splitk & 7) == 0 ? dim3(tiles * splitk, 1, 1) : dim3(tiles, 1, splitk > 1 ? splitk : 1);
}

// the bf16 x 6 weight-gradient product: producer / consumer kernel (512 threads, 96 KB of LDS), or the per-wave kernel of round 5 on request
static int launch_tn_x6(int mode, int tiles, int splitk, hipStream_t st, int M, int N, int K, float alpha, const float* A, long lda, const float* B, long ldb,
                        float beta, float* C, long ldc, const float* bias, int klen, float* slabs, const float* A2, long lda2, int msplit) {
    // mode: FN_GEMM_X6_PERWAVE = the round-5 kernel, FN_GEMM_X6_WIDE = 128 x 256 tiles, 0 = 128 x 128 tiles (producer / consumer kernels)
    if (mode & FN_GEMM_X6_PERWAVE) {
        hipLaunchKernelGGL(gemm_tn_x6_kernel, tn_grid(tiles, splitk), dim3(NT), 0, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, klen, slabs, A2, lda2, msplit);
        return FN_OK;
    }
    const bool wide = (mode & FN_GEMM_X6_WIDE) != 0;
    const size_t lds = wide ? (size_t)2 * X6V_STAGE * 16 : (size_t)X6W_STAGES * X6W_STAGE * 16;
    const void* fn = wide ? reinterpret_cast<const void*>(gemm_tn_x6v_kernel) : reinterpret_cast<const void*>(gemm_tn_x6w_kernel);
    static std::atomic<bool> attr_set[2][32];      // write-once per device and kernel; setting the attribute twice is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    if (!attr_set[wide][dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set[wide][dev].store(true, std::memory_order_release);
    }
    // 1-D launches (K ranges dealt to the XCDs): one workgroup per CU walking its (tile, K range) items instead of one workgroup per item - the next
    // item's first blocks are requested and cut while the consumers store the finished one (FN_GEMM_X6_PERTILE: one workgroup per item, A/B and tests)
    const int wtiles = wide ? ((M + 127) / 128) * ((N + 255) / 256) : tiles;
    dim3 grid = tn_grid(wtiles, splitk);
    const int cus = gemm_cu_count();
    if (grid.y == 1 && grid.z == 1 && slabs != nullptr && !(mode & FN_GEMM_X6_PERTILE) && cus >= 8 && (cus & 7) == 0 && (int)grid.x > cus) grid.x = cus;
    if (wide) {
        hipLaunchKernelGGL(gemm_tn_x6v_kernel, grid, dim3(X6W_NT), lds, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, klen, slabs, A2, lda2, msplit);
    } else {
        hipLaunchKernelGGL(gemm_tn_x6w_kernel, grid, dim3(X6W_NT), lds, st, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, klen, slabs, A2, lda2, msplit);
    }
    return FN_OK;
}

size_t fn_gemm_ws_bytes(int M, int N, int splitk) { return splitk > 1 ? (size_t)splitk * M * N * sizeof(float) : 0; }

int fn_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, float alpha, const float* A, int lda, const float* B, int ldb,
                float beta, float* C, int ldc, const float* bias, int splitk, float* ws, size_t ws_bytes, void* stream) {
    if (!A || !B || !C) return FN_E_NULL;
    if (M <= 0 || N <= 0 || K <= 0 || lda <= 0 || ldb <= 0 || ldc < N) return FN_E_SHAPE;
    const bool lean = (splitk & FN_GEMM_LEAN) != 0;
    const bool x6 = (splitk & FN_GEMM_BF16X6) != 0;
    const int x6_mode = splitk & (FN_GEMM_X6_PERWAVE | FN_GEMM_X6_WIDE | FN_GEMM_X6_PERTILE);
    splitk &= ~(FN_GEMM_LEAN | FN_GEMM_BF16X6 | FN_GEMM_X6_PERWAVE | FN_GEMM_X6_WIDE | FN_GEMM_X6_PERTILE);
    if (splitk > 1 && (!ws || ws_bytes < fn_gemm_ws_bytes(M, N, splitk))) return FN_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!a_kmajor && !b_kmajor && (lda % 4) == 0 && (ldb % 4) == 0 && lda >= 4 && ldb >= 4 &&
        (((((uintptr_t)A) | ((uintptr_t)B)) & 15) == 0)) {
        int klen = K;
        if (splitk > 1) {
            // K ranges of whole 32-k blocks for the bf16 x 6 kernel (no fp32-MFMA tail inside a range), of 4 rows otherwise
            const int gran = x6 ? 32 : 4;
            klen = ((K + splitk - 1) / splitk + gran - 1) / gran * gran;
            splitk = (K + klen - 1) / klen;
        }
        const int ntm = (M + 127) / 128, ntn = (N + 127) / 128;
        float* slabs = splitk > 1 ? ws : nullptr;
        if (x6) {
            const int rc = launch_tn_x6(x6_mode, ntm * ntn, splitk, st, M, N, K, alpha, A, (long)lda, B, (long)ldb, beta, C, (long)ldc, bias, klen, slabs,
                                        (const float*)nullptr, 0L, 0);
            if (rc != FN_OK) return rc;
        } else {
            hipLaunchKernelGGL(lean ? gemm_tn_lean_kernel : gemm_tn_kernel, tn_grid(ntm * ntn, splitk), dim3(NT), 0, st, M, N, K, alpha, A,
                               (long)lda, B, (long)ldb, beta, C, (long)ldc, bias, klen, slabs, (const float*)nullptr, 0L, 0);
        }
        FN_CHECK_LAUNCH();
        if (splitk > 1) {
            launch_slab_reduce(st, slabs, splitk, M, N, alpha, beta, C, (long)ldc, bias);
            FN_CHECK_LAUNCH();
        }
        return FN_OK;
    }
    // bf16 x 6, K-contiguous operands, whole 128 x 128 tiles and 32-k blocks, no split: the producer / consumer kernel
    if (x6 && a_kmajor && b_kmajor && splitk <= 1 && (M % 128) == 0 && (N % 128) == 0 && (K % 32) == 0 && K >= 128 && (lda % 4) == 0 && (ldb % 4) == 0 &&
        (ldc % 4) == 0 && (((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0) && (long)(M / 128) * (N / 128) >= 128) {
        const size_t lds = (size_t)X6W_STAGES * X6W_STAGE * 16;
        static std::atomic<bool> attr_set[32];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
        if (!attr_set[dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_x6w_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            attr_set[dev].store(true, std::memory_order_release);
        }
        // one workgroup per CU (144 KB of LDS each) walking its tiles, unless asked for one workgroup per tile (FN_GEMM_X6_PERTILE) or the CU count is not a
        // multiple of the 8 XCDs
        const int ntiles = (M / 128) * (N / 128), cus = gemm_cu_count();
        const int grid = (ntiles > cus && cus >= 8 && (cus & 7) == 0 && !(x6_mode & FN_GEMM_X6_PERTILE)) ? cus : ntiles;
        hipLaunchKernelGGL(gemm_nt_x6w_kernel, dim3(grid), dim3(X6W_NT), lds, st, M, N, K, alpha, A, (long)lda, B, (long)ldb, beta, C, (long)ldc, bias);
        FN_CHECK_LAUNCH();
        return FN_OK;
    }
    // K-contiguous operands, whole 128 x 128 tiles that fill the chip, short K, no split: the LDS-free kernel
    if (a_kmajor && b_kmajor && splitk <= 1 && (M % 128) == 0 && (N % 128) == 0 && (K % 16) == 0 && K >= 64 && (lda % 4) == 0 && (ldb % 4) == 0 &&
        (ldc % 4) == 0 && (((((uintptr_t)A) | ((uintptr_t)B) | ((uintptr_t)C) | ((uintptr_t)bias)) & 15) == 0) &&
        (long)M * lda < (1L << 30) && (long)N * ldb < (1L << 30) && (long)(M / 128) * (N / 128) >= 256) {
        if (lean)       // <= 128 registers, no LDS: one of its wavefronts fits a SIMD beside a wavefront of the decoder-shaped forward scans (377 registers)
            hipLaunchKernelGGL((gemm_nt_direct_kernel<1, 4>), dim3((M / 128) * (N / 128)), dim3(NT), 0, st, M, N, K, alpha, A, (long)lda, B, (long)ldb, beta, C,
                               (long)ldc, bias);
        else
            hipLaunchKernelGGL((gemm_nt_direct_kernel<4, 2>), dim3((M / 128) * (N / 128)), dim3(NT), 0, st, M, N, K, alpha, A, (long)lda, B, (long)ldb, beta, C,
                               (long)ldc, bias);
        FN_CHECK_LAUNCH();
        return FN_OK;
    }
    // skinny outputs (the attribute decoders' output layers: 16384 x 3 / 16 x 512): 64 x 16 tiles, 12.8 KB of LDS - a workgroup fits a CU beside one of the
    // bf16 x 6 producer / consumer kernels (144 of 160 KB), so the loss chain of the step runs beside the dhx1 product instead of behind it; same k order
    if (N <= 16 && a_kmajor && b_kmajor && splitk <= 1 && M >= 64)
        return launch_gemm<64, 16, 16, 4, 1>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, splitk, ws, st);
    const long tiles128 = (long)((M + 127) / 128) * ((N + 127) / 128);
    // 128 x 128 tiles only when they fill the chip (one workgroup per CU); below that the 64 x 64 kernel's 4x more workgroups win
    // (decode at 2048 rows: 192 big tiles -> 55 us, 768 small ones -> 40 us per W_ih2 projection)
    if (tiles128 * (splitk > 1 ? splitk : 1) >= 256)
        return launch_gemm<128, 128, 32, 2, 2>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, splitk, ws, st);
    // 64 x 64 tiles: K tiles of 32 when K allows it - one barrier per 32 MFMAs of a wave instead of per 16 (the decode's output layer,
    // 2048 x 342 x 512: 20.5 -> see profiles); same k order
    if (splitk <= 1 && (K % 32) == 0 && K >= 128) {
        // ... and 32 x 64 tiles when even the 64 x 64 ones leave CUs empty (the decode's output layer at 2048 rows: 192 workgroups -> 384;
        // same k order per output element: bit-identical)
        const long tiles64 = (long)((M + 63) / 64) * ((N + 63) / 64);
        if (a_kmajor && b_kmajor && tiles64 < 240 && M >= 64)
            return launch_gemm<32, 64, 32, 2, 2>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, splitk, ws, st);
        return launch_gemm<64, 64, 32, 2, 2>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, splitk, ws, st);
    }
    return launch_gemm<64, 64, 16, 2, 2>(a_kmajor, b_kmajor, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, bias, splitk, ws, st);
}

size_t fn_gru_dwhh_ws_bytes(int H, int splitk) { return fn_gemm_ws_bytes(3 * H, H, splitk); }

int fn_gru_dwhh_f32(const float* dgx, const float* dghn, const float* hprev, int64_t rows, int H, float beta, float* dW, int splitk,
                    float* ws, size_t ws_bytes, void* stream) {
    if (!dgx || !dghn || !hprev || !dW) return FN_E_NULL;
    if (rows <= 0 || rows > 0x7fffffff || H <= 0) return FN_E_SHAPE;
    const bool lean = (splitk & FN_GEMM_LEAN) != 0;
    const bool x6 = (splitk & FN_GEMM_BF16X6) != 0;
    const int x6_mode = splitk & (FN_GEMM_X6_PERWAVE | FN_GEMM_X6_WIDE | FN_GEMM_X6_PERTILE);
    const int xflags = splitk & (FN_GEMM_BF16X6 | FN_GEMM_X6_PERWAVE | FN_GEMM_X6_WIDE | FN_GEMM_X6_PERTILE);
    splitk &= ~(FN_GEMM_LEAN | FN_GEMM_BF16X6 | FN_GEMM_X6_PERWAVE | FN_GEMM_X6_WIDE | FN_GEMM_X6_PERTILE);
    if (splitk > 1 && (!ws || ws_bytes < fn_gru_dwhh_ws_bytes(H, splitk))) return FN_E_WORKSPACE;
    const int M = 3 * H, N = H, K = (int)rows;
    const bool one_launch = (2 * H) % 128 == 0 && (H % 4) == 0 && (((((uintptr_t)dgx) | ((uintptr_t)dghn) | ((uintptr_t)hprev)) & 15) == 0);
    if (!one_launch) {          // two products: rows [0, 2H) from dgx, rows [2H, 3H) from dghn
        const int fl = splitk | xflags;
        int rc = fn_gemm_f32(0, 0, 2 * H, N, K, 1.0f, dgx, 3 * H, hprev, H, beta, dW, H, nullptr, fl, ws, ws_bytes, stream);
        if (rc != FN_OK) return rc;
        return fn_gemm_f32(0, 0, H, N, K, 1.0f, dghn, H, hprev, H, beta, dW + (size_t)2 * H * H, H, nullptr, fl, ws, ws_bytes, stream);
    }
    hipStream_t st = (hipStream_t)stream;
    int klen = K;
    if (splitk > 1) {
        const int gran = x6 ? 32 : 4;
        klen = ((K + splitk - 1) / splitk + gran - 1) / gran * gran;
        splitk = (K + klen - 1) / klen;
    }
    const int ntm = (M + 127) / 128, ntn = (N + 127) / 128;
    float* slabs = splitk > 1 ? ws : nullptr;
    if (x6) {
        const int rc = launch_tn_x6(x6_mode, ntm * ntn, splitk, st, M, N, K, 1.0f, dgx, (long)3 * H, hprev, (long)H, beta, dW, (long)H, (const float*)nullptr, klen,
                                    slabs, dghn, (long)H, 2 * H);
        if (rc != FN_OK) return rc;
    } else {
        hipLaunchKernelGGL(lean ? gemm_tn_lean_kernel : gemm_tn_kernel, tn_grid(ntm * ntn, splitk), dim3(NT), 0, st, M, N, K, 1.0f, dgx,
                           (long)3 * H, hprev, (long)H, beta, dW, (long)H, (const float*)nullptr, klen, slabs, dghn, (long)H, 2 * H);
    }
    FN_CHECK_LAUNCH();
    if (splitk > 1) {
        launch_slab_reduce(st, slabs, splitk, M, N, 1.0f, beta, dW, (long)H, (const float*)nullptr);
        FN_CHECK_LAUNCH();
    }
    return FN_OK;
}

int fn_transpose_f32(const float* src, int R, int C, int src_ld, float* dst, int dst_ld, void* stream) {
    if (!src || !dst) return FN_E_NULL;
    if (R <= 0 || C <= 0 || src_ld < C || dst_ld < R) return FN_E_SHAPE;
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(32, 8), 0, (hipStream_t)stream, src, R, C,
                       (long)src_ld, dst, (long)dst_ld);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_colsum_multi(const FnColsumJob* jobs, int n_jobs, void* stream) {
    if (!jobs) return FN_E_NULL;
    if (n_jobs <= 0 || n_jobs > FN_COLSUM_MAX_JOBS) return FN_E_COUNT;
    CsArgs a;
    int maxn = 0;
    for (int j = 0; j < n_jobs; ++j) {
        const FnColsumJob& d = jobs[j];
        if (!d.X || !d.out) return FN_E_NULL;
        if (d.M <= 0 || d.N <= 0 || d.ld < d.N) return FN_E_SHAPE;
        a.job[j] = d;
        maxn = d.N > maxn ? d.N : maxn;
    }
    hipLaunchKernelGGL(colsum_multi_kernel, dim3((maxn + 63) / 64, 1, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

static int colsum_chunks(int M) { return M >= 16384 ? 256 : (M >= 4096 ? 64 : (M >= 256 ? 16 : 1)); }
size_t fn_colsum_ws_bytes(int M, int N) { return (size_t)colsum_chunks(M) * N * sizeof(float); }

int fn_colsum_f32(const float* X, int M, int N, int ld, float beta, float* out, float* ws, size_t ws_bytes, void* stream) {
    if (!X || !out || !ws) return FN_E_NULL;
    if (M <= 0 || N <= 0 || ld < N) return FN_E_SHAPE;
    if (ws_bytes < fn_colsum_ws_bytes(M, N)) return FN_E_WORKSPACE;
    const int chunks = colsum_chunks(M), rpc = (M + chunks - 1) / chunks;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((N + 255) / 256, chunks), dim3(256), 0, (hipStream_t)stream, X, M, N, (long)ld, rpc, ws);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(colsum_final_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, chunks, N, beta, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_axpy_f32(int64_t n, float alpha, const float* x, float* y, void* stream) {
    if (!x || !y) return FN_E_NULL;
    if (n <= 0) return FN_E_SHAPE;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(axpy_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (long)n, alpha, x, y);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_sum_f32(const float* x, int64_t n, float scale, float* out, void* stream) {
    if (!x || !out) return FN_E_NULL;
    if (n <= 0) return FN_E_SHAPE;
    hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, (long)n, scale, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
