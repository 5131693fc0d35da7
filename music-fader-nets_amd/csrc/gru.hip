// gru.hip - GRU sequence scans for the GM-VAE path (encoders gmm_model.py:84,89; sub-decoders :109,114;
// global decoder cells :131-136) and their backward (autograd of the same, trainer_gmm.py:249).
//
// One time step of ALL concurrently running scans is ONE launch (a dependent-kernel boundary, ~1.5-1.9 us on
// MI355X, is cheaper than an in-kernel grid barrier and needs no cross-workgroup visibility protocol).
//
// Step kernel structure (forward: h_{p-1} W_hh^T -> gates; backward: dgh W_hh -> dh -> gate backward):
//   * workgroup tile = (16*TM batch rows) x (16 hidden units [x r,z,n] forward / 16*TN units backward);
//   * the 4 waves of a workgroup SPLIT K: wave w owns the 32-wide K chunks w, w+4, ...  and streams its
//     operand fragments global -> VGPR directly in MFMA layout (no LDS staging, no barrier in the K loop).
//     Both operands live in a FRAGMENT-MAJOR image (frag_off): [row tile 16][chunk 32][half][g][i][4 floats],
//     so that the 64 lanes (i = l&15, g = l>>4) of one global_load_dwordx4 read 1 KB of consecutive bytes
//     (a row-major source makes every lane quad touch 4 different cache lines and the kernel TA-bound:
//     measured 3x slower).  The weights are packed once per optimiser step (fn_frag_pack), the recurrent
//     activations are written in this layout by the previous step's epilogue (ping-pong scratch).
//     MFMA step j of a chunk consumes k = 4g + j (j < 4) or 16 + 4g + (j-4) - the same K permutation for
//     A and B, so the product is unchanged;
//   * D chunks are kept in flight per wave (register ring) - the step is latency-bound otherwise;
//   * one LDS exchange + barrier at the end adds the 4 partial accumulators; wave m then owns M-tile m and
//     holds r/z/n of the SAME (row, unit) in the same lane, so the gate epilogue is lane-local;
//   * the saved gates use a private blocked layout in which every wave store / load instruction is one
//     contiguous 256-byte run (see gate_off).
#include <cstdio>
#include <cstdlib>

#include <atomic>

#include "gru_layout.h"
#include "x6w_core.h"

#ifdef FN_TIMING
__device__ unsigned long long fn_dbg[64 * 8];
#define FN_STAMP(k)                                                                                   \
    do {                                                                                              \
        if (blockIdx.x < 64 && threadIdx.x == 0) fn_dbg[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int fn_dbg_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fn_dbg), sizeof(unsigned long long) * 64 * 8);
}
#else
#define FN_STAMP(k)
#endif

namespace {

constexpr int NT = 256;

// ---------------------------------------------------------------------------------------------
// forward step
// ---------------------------------------------------------------------------------------------
struct FwdStep {
    const float* w_frag;   // W_hh, fragment-major
    const float* b_hh;
    const float* b_ih;
    const float* h_prev;   // [B][H] row-major state before this step (epilogue operand); null = zeros
    const float* hf_in;    // the same state, fragment-major (MFMA operand)
    float* hf_out;         // state after this step, fragment-major (next step's operand)
    float* h_out;
    float* gates;          // null = do not save
    const float* gx_dense;
    const float* gx_table;
    const int* idx;        // already offset to column tau; null -> tok_const
    const float* gx_rowbias;
    int idx_ld, tok_const;
    int B, H;
    int tile0, ntm;        // first virtual tile of this scan, number of row tiles
};
struct FwdArgs {
    FwdStep s[FN_MAX_SCANS];
    int n, total;
};

// NS = number of scans covered by this launch (a distinct kernel symbol per phase: 4 = the encoder step,
// 3 = decoder layer 1 + both sub-decoders, 1 = a single scan), so per-phase durations show up separately in rocprof.
template <int NS, int TM, int D>
__global__ __launch_bounds__(NT) void gru_fwd_step_kernel(const FwdArgs args) {
    __shared__ __attribute__((aligned(16))) float red[4 * TM * 3 * 256];
    const int v = fn_xcd_remap(blockIdx.x, args.total);
    int si = 0;
#pragma unroll
    for (int k = 1; k < NS; ++k)
        if (k < args.n && v >= args.s[k].tile0) si = k;
    const FwdStep& S = args.s[si];
    const int local = v - S.tile0;
    const int tn = local / S.ntm, tm = local % S.ntm;     // row tiles fastest: neighbours share weight rows
    const int m0 = tm * (16 * TM), hh0 = tn * 16;
    const int B = S.B, H = S.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nrt = (B + 15) >> 4;

    FN_STAMP(0);
    const int jj = hh0 + li;
    const bool has_k = S.hf_in != nullptr && S.h_prev != nullptr;
    const int nk = H >> 5;
    const int nkw = (has_k && wave < nk) ? (nk - wave + 3) >> 2 : 0;        // this wave's chunks: wave, wave+4, ...

    // (1) token ids of the rows this wave finishes (a compiler-visible load; the table rows depend on it)
    int tok[4] = {0, 0, 0, 0};
    if (wave < TM && S.gx_table) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = min(m0 + wave * 16 + lg * 4 + i, B - 1);
            tok[i] = S.idx ? S.idx[(long)b * S.idx_ld] : S.tok_const;
        }
    }

    // (2) epilogue operands of the M-tile this wave will finish: LOADED now (their latency hides under the K loop), only
    //     ADDED in the epilogue (an add here would make hipcc wait for them before the loop)
    float e_tab[4][3], e_den[4][3], e_rb[4][3], hp[4], bh[3], bi[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { bh[q] = 0.f; bi[q] = 0.f; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hp[i] = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) { e_tab[i][q] = 0.f; e_den[i][q] = 0.f; e_rb[i][q] = 0.f; }
    }
    if (wave < TM) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bh[q] = S.b_hh[q * H + jj];
            if (S.b_ih) bi[q] = S.b_ih[q * H + jj];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = min(m0 + wave * 16 + lg * 4 + i, B - 1);
            if (S.gx_dense) {
                const float* row = S.gx_dense + (long)b * 3 * H;
#pragma unroll
                for (int q = 0; q < 3; ++q) e_den[i][q] = row[q * H + jj];
            }
            if (S.gx_table) {
                const float* row = S.gx_table + (long)tok[i] * 3 * H;
#pragma unroll
                for (int q = 0; q < 3; ++q) e_tab[i][q] = row[q * H + jj];
            }
            if (S.gx_rowbias) {
                const float* row = S.gx_rowbias + (long)b * 3 * H;
#pragma unroll
                for (int q = 0; q < 3; ++q) e_rb[i][q] = row[q * H + jj];
            }
            if (S.h_prev) hp[i] = S.h_prev[(long)b * H + jj];
        }
    }

    // (3) first D operand chunks of the K loop (asm loads, counted by us - see fn_gld4_asm).  They are issued AFTER the
    //     compiler-visible loads above: vmcnt is one in-order counter, so older outstanding loads only make our counted
    //     waits stricter, never wrong.
    const float* ap[TM];
    const float* bp[3];
#pragma unroll
    for (int m = 0; m < TM; ++m) ap[m] = S.hf_in + (long)min(tm * TM + m, nrt - 1) * nk * 512 + lane * 4;
#pragma unroll
    for (int n = 0; n < 3; ++n) bp[n] = S.w_frag + (long)(n * (H >> 4) + tn) * nk * 512 + lane * 4;
    f32x4 fa[D][TM][2], fb[D][3][2];
    constexpr int NL = 2 * (TM + 3);                                       // loads per chunk
    auto load = [&](int set, int it) {
        const int k0 = (wave + 4 * min(it, nkw - 1)) * 512;                // one chunk = 2 halves x 64 lanes x 4 floats
#pragma unroll
        for (int m = 0; m < TM; ++m) { fn_gld4_asm(fa[set][m][0], ap[m] + k0); fn_gld4_asm(fa[set][m][1], ap[m] + k0 + 256); }
#pragma unroll
        for (int n = 0; n < 3; ++n) { fn_gld4_asm(fb[set][n][0], bp[n] + k0); fn_gld4_asm(fb[set][n][1], bp[n] + k0 + 256); }
    };
    if (nkw > 0) {
#pragma unroll
        for (int s = 0; s < D; ++s) load(s, s);          // chunk index is clamped: always a legal (possibly repeated) chunk
    }

    f32x4 acc[TM][3];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    FN_STAMP(1);
    // (4) K loop: branch-free steady state, MFMAs read the ring registers in place, load(u) refills them right behind
    auto mma = [&](int set) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(fa[set][m][j >> 2], j & 3), f4at(fb[set][n][j >> 2], j & 3),
                                                                     acc[m][n], 0, 0, 0);
    };
    if (nkw > 0) {
        const int nmain = nkw / D * D;
        for (int base = 0; base < nmain; base += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                fn_wait_vm<NL * (D - 1)>();
                mma(u);
                load(u, base + u + D);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        fn_wait_vm<0>();
#pragma unroll
        for (int u = 0; u < D; ++u)
            if (nmain + u < nkw) mma(u);                 // leftover chunks are already in the ring
#pragma unroll
        for (int u = 0; u < D; ++u) {                    // every ring register stays allocated until its load has landed
#pragma unroll
            for (int m = 0; m < TM; ++m) { fn_keep(fa[u][m][0]); fn_keep(fa[u][m][1]); }
#pragma unroll
            for (int n = 0; n < 3; ++n) { fn_keep(fb[u][n][0]); fn_keep(fb[u][n][1]); }
        }
    }
    FN_STAMP(2);
    if (has_k) {
        // ---- add the 4 K-partials: wave m ends up with M-tile m --------------------------------------------
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                *reinterpret_cast<f32x4*>(red + ((wave * TM + m) * 3 + n) * 256 + lane * 4) = acc[m][n];
        __syncthreads();
    }
    FN_STAMP(3);
    if (wave >= TM) return;
    f32x4 r3[3];
#pragma unroll
    for (int n = 0; n < 3; ++n) {
        r3[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (has_k) {
#pragma unroll
            for (int w = 0; w < 4; ++w) r3[n] += *reinterpret_cast<const f32x4*>(red + ((w * TM + wave) * 3 + n) * 256 + lane * 4);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = m0 + wave * 16 + lg * 4 + i;
        if (b >= B) continue;
        const float ghr = r3[0][i] + bh[0], ghz = r3[1][i] + bh[1], ghn = r3[2][i] + bh[2];
        const float gxr = ((bi[0] + e_den[i][0]) + e_tab[i][0]) + e_rb[i][0];
        const float gxz = ((bi[1] + e_den[i][1]) + e_tab[i][1]) + e_rb[i][1];
        const float gxn = ((bi[2] + e_den[i][2]) + e_tab[i][2]) + e_rb[i][2];
        const float r = fn_sigmoid(gxr + ghr);
        const float z = fn_sigmoid(gxz + ghz);
        const float n = fn_tanh(gxn + r * ghn);
        const float h = (1.0f - z) * n + z * hp[i];
        S.h_out[(long)b * H + jj] = h;
        if (S.hf_out) S.hf_out[frag_off(b, jj, H >> 5)] = h;
        if (S.gates) {
            float* g = S.gates;
            g[gate_off(b, 0, jj, nrt)] = r;
            g[gate_off(b, 1, jj, nrt)] = z;
            g[gate_off(b, 2, jj, nrt)] = n;
            g[gate_off(b, 3, jj, nrt)] = ghn;
        }
    }
    FN_STAMP(4);
}

// ---------------------------------------------------------------------------------------------
// backward step:  dh_{q} = [dgh_{q+1} W_hh] + dh_in + dh_ext ; then the gate backward of step q
// ---------------------------------------------------------------------------------------------
struct BwdStep {
    const float* wt_frag;  // W_hh^T ([H][3H]), fragment-major
    const float* df_in;    // [dr | dz | dn*r] pre-activation gradients of step q+1 ([B][3H]), fragment-major; null = no GEMM
    float* df_out;         // the same for step q (next iteration's operand)
    const float* dh_in;    // [B][H] carried dh*z (or dh_last on the first iteration); may alias dhz_out
    const float* dh_ext;   // [B][H] or null
    const float* gates_q;  // null = no gate backward (final iteration: write dh0_out)
    const float* hprev_q;  // h before step q, null = zeros
    float* dgx_q;
    float* dghn_q;
    float* dhz_out;
    float* rowsum;
    float* rowsum_n;
    float* dh0_out;
    int B, H;
    int tile0, ntm;
};
struct BwdArgs {
    BwdStep s[FN_MAX_SCANS];
    int n, total;
};

template <int NS, int TM, int TN, int D>
__global__ __launch_bounds__(NT) void gru_bwd_step_kernel(const BwdArgs args) {
    __shared__ __attribute__((aligned(16))) float red[4 * TM * TN * 256];
    const int v = fn_xcd_remap(blockIdx.x, args.total);
    int si = 0;
#pragma unroll
    for (int k = 1; k < NS; ++k)
        if (k < args.n && v >= args.s[k].tile0) si = k;
    const BwdStep& S = args.s[si];
    const int local = v - S.tile0;
    const int tn = local / S.ntm, tm = local % S.ntm;
    const int m0 = tm * (16 * TM), n0 = tn * (16 * TN);
    const int B = S.B, H = S.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int nrt = (B + 15) >> 4;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int nk = (3 * H) >> 5;
    const int nkw = (S.df_in && wave < nk) ? (nk - wave + 3) >> 2 : 0;
    if (S.df_in) {
        const float *ap[TM], *bp[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) ap[m] = S.df_in + (long)min(tm * TM + m, nrt - 1) * nk * 512 + lane * 4;
#pragma unroll
        for (int n = 0; n < TN; ++n) bp[n] = S.wt_frag + (long)min(tn * TN + n, (H >> 4) - 1) * nk * 512 + lane * 4;
        f32x4 fa[D][TM][2], fb[D][TN][2];
        constexpr int NL = 2 * (TM + TN);
        auto load = [&](int set, int it) {
            const int k0 = (wave + 4 * min(it, nkw - 1)) * 512;
#pragma unroll
            for (int m = 0; m < TM; ++m) { fn_gld4_asm(fa[set][m][0], ap[m] + k0); fn_gld4_asm(fa[set][m][1], ap[m] + k0 + 256); }
#pragma unroll
            for (int n = 0; n < TN; ++n) { fn_gld4_asm(fb[set][n][0], bp[n] + k0); fn_gld4_asm(fb[set][n][1], bp[n] + k0 + 256); }
        };
        auto mma = [&](int set) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(fa[set][m][j >> 2], j & 3), f4at(fb[set][n][j >> 2], j & 3),
                                                                         acc[m][n], 0, 0, 0);
        };
        if (nkw > 0) {
#pragma unroll
            for (int s = 0; s < D; ++s) load(s, s);
            const int nmain = nkw / D * D;
            for (int base = 0; base < nmain; base += D) {
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    fn_wait_vm<NL * (D - 1)>();
                    mma(u);
                    load(u, base + u + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fn_wait_vm<0>();
#pragma unroll
            for (int u = 0; u < D; ++u)
                if (nmain + u < nkw) mma(u);
#pragma unroll
            for (int u = 0; u < D; ++u) {
#pragma unroll
                for (int m = 0; m < TM; ++m) { fn_keep(fa[u][m][0]); fn_keep(fa[u][m][1]); }
#pragma unroll
                for (int n = 0; n < TN; ++n) { fn_keep(fb[u][n][0]); fn_keep(fb[u][n][1]); }
            }
        }
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                *reinterpret_cast<f32x4*>(red + ((wave * TM + m) * TN + n) * 256 + lane * 4) = acc[m][n];
        __syncthreads();
    }
    if (wave >= TM) return;
#pragma unroll
    for (int t2 = 0; t2 < TN; ++t2) {
        f32x4 a4 = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (S.df_in) {
#pragma unroll
            for (int w = 0; w < 4; ++w) a4 += *reinterpret_cast<const f32x4*>(red + ((w * TM + wave) * TN + t2) * 256 + lane * 4);
        }
        const int jj = n0 + t2 * 16 + li;
        if (jj >= H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = m0 + wave * 16 + lg * 4 + i;
            if (b >= B) continue;
            const long o = (long)b * H + jj;
            float dh = a4[i];
            if (S.dh_in) dh += S.dh_in[o];
            if (S.dh_ext) dh += S.dh_ext[o];
            if (!S.gates_q) {
                S.dh0_out[o] = dh;
                continue;
            }
            const float* g = S.gates_q;
            const float r = g[gate_off(b, 0, jj, nrt)], z = g[gate_off(b, 1, jj, nrt)], n = g[gate_off(b, 2, jj, nrt)],
                        hn = g[gate_off(b, 3, jj, nrt)];
            const float hp = S.hprev_q ? S.hprev_q[o] : 0.f;
            const float dn = dh * (1.0f - z);
            const float dz = dh * (hp - n);
            const float dnp = dn * (1.0f - n * n);
            const float dr = dnp * hn;
            const float dzp = dz * z * (1.0f - z);
            const float drp = dr * r * (1.0f - r);
            float* dg = S.dgx_q + (long)b * 3 * H;
            dg[jj] = drp; dg[H + jj] = dzp; dg[2 * H + jj] = dnp;
            S.dghn_q[o] = dnp * r;
            S.dhz_out[o] = dh * z;
            if (S.df_out) {
                const int NC3 = (3 * H) >> 5;
                S.df_out[frag_off(b, jj, NC3)] = drp;
                S.df_out[frag_off(b, H + jj, NC3)] = dzp;
                S.df_out[frag_off(b, 2 * H + jj, NC3)] = dnp * r;
            }
            if (S.rowsum) {
                float* rs = S.rowsum + (long)b * 3 * H;
                rs[jj] += drp; rs[H + jj] += dzp; rs[2 * H + jj] += dnp;
            }
            if (S.rowsum_n) S.rowsum_n[o] += dnp * r;
        }
    }
}

// row-major [rows][K] (leading dimension ld) -> fragment-major image with rows padded to a multiple of 16 (zero filled)
__global__ void frag_pack_kernel(const float* __restrict__ src, int rows, int K, long ld, float* __restrict__ dst) {
    const int rows16 = (rows + 15) & ~15;
    const long total = (long)rows16 * (K >> 2);
    const bool vec = ((((uintptr_t)src) & 15) == 0) && ((ld & 3) == 0);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int row = (int)(i / (K >> 2)), k = (int)(i % (K >> 2)) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows) {
            const float* p = src + (long)row * ld + k;
            if (vec) v = *reinterpret_cast<const float4*>(p);
            else { v.x = p[0]; v.y = p[1]; v.z = p[2]; v.w = p[3]; }
        }
        *reinterpret_cast<float4*>(dst + frag_off(row, k, K >> 5)) = v;
    }
}

// row-major [rows][K] -> bf16 TRIPLE image for the bf16 x 6 kernels (gru_persist.hip: gru_fwd_x6_kernel): [row tile 16][k block 32][piece 3][64 lanes][8 bf16],
// lane (i, g) = row 16 t + i, k values 32 b + 8 g .. + 7; piece 0 / 1 / 2 = hi / mid / lo with hi + mid + lo == src exactly (rounded pieces: fn_rn16);
// rows padded to a multiple of 16 with zeros.  One thread per (row, 8 k).
// item i = (row, 8 k) of the triple image of a [rows][K] matrix whose element (row, k) lives at src[row * sr + k * sk]
// ROWFAST: consecutive items walk the ROWS of one 8-k group (the transposed source, sr == 1: a wavefront's loads are 256 contiguous bytes per k
// instead of 64 lines of 4 bytes)
template <bool ROWFAST = false>
FN_DEVINL void frag3_item(const float* __restrict__ src, int rows, int K, long sr, long sk, unsigned* __restrict__ dst, long i) {
    const int nb = K >> 5;
    const int rows16 = (rows + 15) & ~15;
    const int row = ROWFAST ? (int)(i % rows16) : (int)(i / (K >> 3)), k8 = ROWFAST ? (int)(i / rows16) : (int)(i % (K >> 3));
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = row < rows ? src[(long)row * sr + (long)(8 * k8 + j) * sk] : 0.f;
    unsigned* o = dst + ((((long)(row >> 4) * nb + (k8 >> 2)) * 3) * 64 + (row & 15) + 16 * (k8 & 3)) * 4;
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float hi[2], mi[2], r1[2], r2[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float v = x[2 * j + e];
            hi[e] = fn_rn16(v);
            r1[e] = v - hi[e];
            mi[e] = fn_rn16(r1[e]);
            r2[e] = r1[e] - mi[e];
        }
        h[j] = (__float_as_uint(hi[0]) >> 16) | (__float_as_uint(hi[1]) & 0xffff0000u);
        m[j] = (__float_as_uint(mi[0]) >> 16) | (__float_as_uint(mi[1]) & 0xffff0000u);
        l[j] = (__float_as_uint(r2[0]) >> 16) | (__float_as_uint(r2[1]) & 0xffff0000u);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { o[j] = h[j]; o[256 + j] = m[j]; o[512 + j] = l[j]; }
}

__global__ void frag3_pack_kernel(const float* __restrict__ src, int rows, int K, long ld, unsigned* __restrict__ dst) {
    const long total = (long)((rows + 15) & ~15) * (K >> 3);
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) frag3_item(src, rows, K, ld, 1, dst, i);
}

// ---- all weight images of one optimiser step in ONE launch (fn_weight_images) -------------------------------------------------------
// job kinds: 0 = dst [C][R] = src^T (src [R][C], leading dimension ld): the one-hot column table W_ih[:, :V]^T
//            1 = fragment-major image of src [rows = R][K = C]                           (forward scans: W_hh, W_ih2, W_out)
//            2 = fragment-major image of src^T, i.e. of the [rows = C][K = R] matrix    (backward scans: W_hh^T) - no intermediate
//            3 = bf16 triple image (fn_frag3_pack layout) of src [rows = R][K = C]       (bf16 x 6 forward scans)
//            4 = bf16 triple image of src^T, the [rows = C][K = R] matrix                (bf16 x 6 backward scans: W_hh^T)
//            5 = dst [R][C] dense = src [R][C] (leading dimension ld): aligned image of a column slice (W_ih[:, V:], the z part of a cell's input matrix)
// The refresh after every Adam update used to be ~40 dependent launches of a few microseconds of work each (0.5 ms of the step).
constexpr int WI_MAX_JOBS = 56;
struct WiJob {
    const float* src;
    float* dst;
    int R, C, ld, kind;
};
struct WiArgs {
    WiJob job[WI_MAX_JOBS];
};

__global__ __launch_bounds__(256) void weight_images_kernel(const WiArgs a) {
    const WiJob& J = a.job[blockIdx.y];
    if (J.kind == 0) {
        // LDS-tiled transpose, 32 x 32 tiles, grid-stride over tiles
        __shared__ float t[32][33];
        const int tr = (J.R + 31) / 32, tc = (J.C + 31) / 32;
        const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
        for (int tile = blockIdx.x; tile < tr * tc; tile += gridDim.x) {
            const int r0 = (tile / tc) * 32, c0 = (tile % tc) * 32;
            for (int j = ty; j < 32; j += 8) {
                const int r = r0 + j, c = c0 + tx;
                t[j][tx] = (r < J.R && c < J.C) ? J.src[(long)r * J.ld + c] : 0.f;
            }
            __syncthreads();
            for (int j = ty; j < 32; j += 8) {
                const int c = c0 + j, r = r0 + tx;
                if (r < J.R && c < J.C) J.dst[(long)c * J.R + r] = t[tx][j];
            }
            __syncthreads();
        }
        return;
    }
    if (J.kind == 5) {
        // dense copy dst [R][C] of a column slice src [R][C] (leading dimension ld): a 16-byte aligned operand for the products that read the slice
        const long total = (long)J.R * J.C;
        for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) J.dst[i] = J.src[(i / J.C) * J.ld + (i % J.C)];
        return;
    }
    if (J.kind >= 3) {
        const int rows = J.kind == 3 ? J.R : J.C, K = J.kind == 3 ? J.C : J.R;
        const long total = (long)((rows + 15) & ~15) * (K >> 3);
        if (J.kind == 3)
            for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L)
                frag3_item<false>(J.src, rows, K, (long)J.ld, 1L, reinterpret_cast<unsigned*>(J.dst), i);
        else
            for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L)
                frag3_item<true>(J.src, rows, K, 1L, (long)J.ld, reinterpret_cast<unsigned*>(J.dst), i);
        return;
    }
    const int rows = J.kind == 1 ? J.R : J.C, K = J.kind == 1 ? J.C : J.R;
    const int rows16 = (rows + 15) & ~15, NC = K >> 5;
    const long total4 = (long)rows16 * (K >> 2);          // one item = 4 consecutive k of one row = 16 contiguous bytes of the image
    for (long i = blockIdx.x * 256L + threadIdx.x; i < total4; i += (long)gridDim.x * 256L) {
        // walk the IMAGE linearly (coalesced 16-byte stores): invert frag_off for the item's first float
        const long o = i * 4;
        const int lane = (int)((o >> 2) & 63), half = (int)((o >> 8) & 1);
        const long blk = o >> 9;                            // (row >> 4) * NC + (k >> 5)
        const int row = (int)(blk / NC) * 16 + (lane & 15);
        const int k = (int)(blk % NC) * 32 + half * 16 + (lane >> 4) * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows) {
            if (J.kind == 1) {
                const float* p = J.src + (long)row * J.ld + k;
                v = make_float4(p[0], p[1], p[2], p[3]);
            } else {
                const float* p = J.src + (long)k * J.ld + row;
                v = make_float4(p[0], p[J.ld], p[2L * J.ld], p[3L * J.ld]);
            }
        }
        *reinterpret_cast<float4*>(J.dst + o) = v;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// One GRU cell step for a LARGE batch (eval-mode greedy decode of thousands of rows, gmm_model.py:131-136) as a staged GEMM with the
// gate non-linearities in its epilogue.  The per-step scan kernel above splits K over the 4 waves of a 64-row tile and re-reads
// its W_hh slice and state tile from L2 with 14 flop per byte (70 TFLOP/s at 2048 rows), and layer 2 needed a separate W_ih
// projection launch.  Here a workgroup owns 128 rows x 32 hidden units: the 96 weight rows of the three gates are gathered by the
// row map, the x part (x W_ih^T, optional) and the h part (h W_hh^T) run through the same LDS-staged K loop into separate
// accumulators (the n gate needs them apart: tanh(gi_n + r * gh_n)), and every lane ends up holding r, z, n of its (row, unit)
// pairs - no reduction, no exchange.
struct RowsGate {                                    // local B-tile row r = 32 gate + unit -> weight row gate * H + u0 + unit (shifts only: it runs in every operand load)
    int u0, H;
    FN_DEVINL bool valid(int r) const { return true; }
    FN_DEVINL long clamped(int r) const { return (long)(r >> 5) * H + u0 + (r & 31); }
    FN_DEVINL bool all_valid(int rows) const { return true; }
};

struct CellArgs {
    const float* x; long ldx; int K1;                // dense input [B][K1] (NULL: none)
    const float* w_ih; long ldw_ih;                  // [3H][K1]
    const float* gx_table; const int* idx; long idx_ld; int tok_const;    // optional token row gx_table[tok][3H]; idx NULL -> tok_const
    const float* gx_rowbias;                         // [B][3H] or NULL
    const float* h_prev; long ldh;                   // [B][H]
    const float* w_hh; long ldw_hh;                  // [3H][H]
    const float* b_ih; const float* b_hh;            // [3H] (b_ih may be NULL)
    float* h_out; long ldo;
    int B, H;
    const unsigned long long* best; int best_v;      // optional packed argmax words of the previous token (fn_out_argmax_f32)
    FN_DEVINL int token(int rc) const {
        return best ? best_v - 1 - (int)(unsigned)(best[rc] & 0xffffffffull) : (idx ? idx[(long)rc * idx_ld] : tok_const);
    }
};

constexpr int GC_BN = 96, GC_BK = 32;
// BM rows x 32 units per workgroup, waves WM x WN: WN = 2 gives each wave one 16-unit tile (all three gates of it)
template <int BM, int WM, int WN>
__global__ __launch_bounds__(NT, 2) void gru_cell_kernel(const CellArgs a) {
    static_assert(WM * WN == 4 && (WN == 1 || WN == 2), "4 waves");
    constexpr int TM = BM / WM / 16, TN = GC_BN / WN / 16, UT = 2 / WN;      // a wave's TN = 3 UT column tiles: tile n = UT * gate + ut, B rows 32 gate + 16 (wn + ut)
    constexpr int BST = WN == 2 ? 32 : 16;             // B-tile rows between consecutive column tiles of a wave
    using SA = Stage<BM, GC_BK, true, NT>;
    using SB = Stage<GC_BN, GC_BK, true, NT>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int nut = a.H >> 5;
    // 4 x 8 super-tiles (row panels x unit tiles) in consecutive virtual ids = on one XCD
    const int v = fn_xcd_remap(blockIdx.x, gridDim.x);
    const int ntm = (a.B + BM - 1) / BM;
    int tm, tu;
    if ((ntm % 4) == 0 && (nut % 8) == 0) {
        const int sb = v >> 5, l = v & 31, sbu = nut >> 3;
        tm = (sb / sbu) * 4 + (l >> 3);
        tu = (sb % sbu) * 8 + (l & 7);
    } else {
        tm = v / nut;
        tu = v % nut;
    }
    const int m0 = tm * BM, u0 = tu * 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int arow0 = wm * (BM / WM), brow0 = WN == 2 ? 16 * wn : 0;
    const RowsPlain ra{m0, a.B};
    const RowsGate rb{u0, a.H};
    f32x4 ax[TM][TN], ah[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n) ax[m][n] = ah[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (a.x) {
        const int nk = (a.K1 + GC_BK - 1) / GC_BK;
        if (SA::can_fast(a.x, a.ldx, ra, a.K1) && SB::can_fast(a.w_ih, a.ldw_ih, rb, a.K1)) {
            auto loadA = [&](int k0, SA& st) { st.load_fast(a.x, a.ldx, ra, k0); };
            auto loadB = [&](int k0, SB& st) { st.load_fast(a.w_ih, a.ldw_ih, rb, k0); };
            fn_kloop<2, TM, TN, GC_BK, SA, SB, BST>(smem, nk, loadA, loadB, arow0, brow0, lane, ax);
        } else {
            auto loadA = [&](int k0, SA& st) { st.load_checked(a.x, a.ldx, ra, k0, a.K1); };
            auto loadB = [&](int k0, SB& st) { st.load_checked(a.w_ih, a.ldw_ih, rb, k0, a.K1); };
            fn_kloop<2, TM, TN, GC_BK, SA, SB, BST>(smem, nk, loadA, loadB, arow0, brow0, lane, ax);
        }
    }
    {
        const int nk = (a.H + GC_BK - 1) / GC_BK;
        if (SA::can_fast(a.h_prev, a.ldh, ra, a.H) && SB::can_fast(a.w_hh, a.ldw_hh, rb, a.H)) {
            auto loadA = [&](int k0, SA& st) { st.load_fast(a.h_prev, a.ldh, ra, k0); };
            auto loadB = [&](int k0, SB& st) { st.load_fast(a.w_hh, a.ldw_hh, rb, k0); };
            fn_kloop<2, TM, TN, GC_BK, SA, SB, BST>(smem, nk, loadA, loadB, arow0, brow0, lane, ah);
        } else {
            auto loadA = [&](int k0, SA& st) { st.load_checked(a.h_prev, a.ldh, ra, k0, a.H); };
            auto loadB = [&](int k0, SB& st) { st.load_checked(a.w_hh, a.ldw_hh, rb, k0, a.H); };
            fn_kloop<2, TM, TN, GC_BK, SA, SB, BST>(smem, nk, loadA, loadB, arow0, brow0, lane, ah);
        }
    }
    // D[row = 16 m + (lane>>4)*4 + i][col = lane&15] of tile n = UT * gate + ut.
    // Every operand of the epilogue is loaded WITHOUT a branch (an absent source reads b_hh and is dropped by a select; rows past the
    // batch read the last row): inside `if (table) {...}` hipcc waits for each load at the end of its block - 96 serialised memory
    // latencies per lane, 35 us of a 56 us layer-1 launch.
    const int cj = lane & 15, rq = (lane >> 4) * 4;
    const bool has_tab = a.gx_table != nullptr, has_rb = a.gx_rowbias != nullptr, has_bi = a.b_ih != nullptr;
    const float* tabp = has_tab ? a.gx_table : a.b_hh;
    const float* rbp = has_rb ? a.gx_rowbias : a.b_hh;
    const float* bip = has_bi ? a.b_ih : a.b_hh;
    const long H3 = 3L * a.H;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = m0 + arow0 + 16 * m + rq + i;
            const int rc = min(row, a.B - 1);
            const int tok = has_tab ? a.token(rc) : 0;
            const long toff = has_tab ? (long)tok * H3 : 0, roff = has_rb ? (long)rc * H3 : 0;
            float tv[UT][3], rv[UT][3], hv[UT];
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int u = u0 + 16 * ((WN == 2 ? wn : 0) + ut) + cj;
                hv[ut] = a.h_prev[(long)rc * a.ldh + u];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    tv[ut][q] = tabp[toff + q * a.H + u];
                    rv[ut][q] = rbp[roff + q * a.H + u];
                }
            }
#pragma unroll
            for (int ut = 0; ut < UT; ++ut) {
                const int u = u0 + 16 * ((WN == 2 ? wn : 0) + ut) + cj;
                float gi[3], gh[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float e = (has_bi ? bip[q * a.H + u] : 0.f) + ax[m][UT * q + ut][i];   // same association as the scan kernels: ((b_ih + x part) + token row) + row constant
                    e = has_tab ? e + tv[ut][q] : e;
                    e = has_rb ? e + rv[ut][q] : e;
                    gi[q] = e;
                    gh[q] = ah[m][UT * q + ut][i] + a.b_hh[q * a.H + u];
                }
                const float r = fn_sigmoid(gi[0] + gh[0]);
                const float z = fn_sigmoid(gi[1] + gh[1]);
                const float n = fn_tanh(gi[2] + r * gh[2]);
                if (row < a.B) a.h_out[(long)row * a.ldo + u] = (1.0f - z) * n + z * hv[ut];
            }
        }
}

// The same cell WITHOUT LDS (the loop of gemm_nt_direct_kernel, gemm.hip): both operands are K-contiguous rows, so lane (i = l & 15, g = l >> 4)
// reads the float4 [row i of a 16-row tile][k0 + 4g .. 4g + 3] of its RT state-row tiles and of the three gate rows of its 16 hidden units
// straight from memory; MFMA j of a 16-k step takes element j of every lane (a permutation of the step's k values, the same for A and B).
// The staged kernel above spends a K = 512 product on 16 barrier-separated tiles (37 / 67 us for the two layers of a 2048-row token
// against 22 / 44 us of MFMA time); here PF steps (RT + 3 loads each) are in flight behind counted waits, nothing else waits.
// One wave = 16 RT rows x 16 units x {r, z, n}: r and z accumulate the x part and the h part in ONE tile, the n gate keeps them apart
// (tanh(gi_n + r * gh_n)); a workgroup = 2 x 2 waves = 32 RT rows x 32 units.  Loads use a scalar base (64 bytes per step) + a fixed
// 32-bit lane offset.  Needs K1, H multiples of 16, 16-byte aligned operands, row strides multiples of 4 floats, spans below 4 GB.
FN_DEVINL void fn_gld4_sb(f32x4& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}

template <int RT, int PF, bool HAS_TAB, bool HAS_RB>
__global__ __launch_bounds__(NT, RT >= 4 ? 1 : 2) void gru_cell_direct_kernel(const CellArgs a) {
    const int nut = a.H >> 5, ntm = (a.B + 32 * RT - 1) / (32 * RT);
    const int v = fn_xcd_remap(blockIdx.x, gridDim.x);
    int tm, tu;
    if ((ntm % 4) == 0 && (nut % 8) == 0) {          // 4 x 8 super-tiles (row panels x unit tiles) in consecutive virtual ids = on one XCD
        const int sb = v >> 5, l = v & 31, sbu = nut >> 3;
        tm = (sb / sbu) * 4 + (l >> 3);
        tu = (sb % sbu) * 8 + (l & 7);
    } else {
        tm = v / nut;
        tu = v % nut;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = tm * 32 * RT + (wave >> 1) * 16 * RT, u0 = tu * 32 + (wave & 1) * 16;
    const int li = lane & 15, lg = lane >> 4;
    f32x4 arz[RT][2], anx[RT], anh[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) arz[m][0] = arz[m][1] = anx[m] = anh[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 fa[PF][RT], fb[PF][3];
    // one K phase: acc_rz += A W[r, z rows]^T, accn += A W[n rows]^T over nks steps of 16 k
    auto phase = [&](const float* A, long lda, const float* W, long ldw, int nks, f32x4 (&accn)[RT]) {
        unsigned oa[RT], ob[3];
#pragma unroll
        for (int m = 0; m < RT; ++m) oa[m] = (unsigned)(((long)min(m0 + 16 * m + li, a.B - 1) * lda + 4 * lg) * 4);
#pragma unroll
        for (int q = 0; q < 3; ++q) ob[q] = (unsigned)(((long)(q * a.H + u0 + li) * ldw + 4 * lg) * 4);
        const float* pa = A;
        const float* pb = W;
        auto load = [&](int set) {
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_gld4_sb(fa[set][m], oa[m], pa);
#pragma unroll
            for (int q = 0; q < 3; ++q) fn_gld4_sb(fb[set][q], ob[q], pb);
            pa += 16;
            pb += 16;
        };
        auto mma = [&](int u) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    arz[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], fb[u][0][j], arz[m][0], 0, 0, 0);
                    arz[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], fb[u][1][j], arz[m][1], 0, 0, 0);
                    accn[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], fb[u][2][j], accn[m], 0, 0, 0);
                }
        };
        constexpr int NL = RT + 3;
        const int nmain = nks / PF * PF;
        if (nmain > 0) {
#pragma unroll
            for (int s = 0; s < PF; ++s) load(s);
            for (int base = 0; base + PF < nmain; base += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    fn_wait_vm<NL * (PF - 1)>();
                    mma(u);
                    load(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (u == 0) fn_wait_vm<NL * (PF - 1)>();
                else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? NL * (PF - 2) : 0)>();
                else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? NL * (PF - 3) : 0)>();
                else fn_wait_vm<0>();
                mma(u);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int s = 0; s < PF; ++s) {
#pragma unroll
                for (int m = 0; m < RT; ++m) fn_keep(fa[s][m]);
#pragma unroll
                for (int q = 0; q < 3; ++q) fn_keep(fb[s][q]);
            }
        }
        for (int ks = nmain; ks < nks; ++ks) {
            load(0);
            fn_wait_vm<0>();
            mma(0);
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_keep(fa[0][m]);
#pragma unroll
            for (int q = 0; q < 3; ++q) fn_keep(fb[0][q]);
        }
    };
    // Epilogue on (row, 4 units) items: the accumulator tiles of row tile m go through a wave-private LDS tile (D layout in, lane
    // (row = l >> 2, units 4 (l & 3) ..) out), so every other operand - old state, token row, row constant - is ONE 16-byte load per
    // gate and item and the new state one 16-byte store (the element-per-lane form issued 7 scalar loads per (row, unit), 112 per lane).
    // All of them are requested HERE, in front of the K loops (plain loads: older than every ring load, so the counted waits cover them);
    // absent sources are not loaded (HAS_TAB / HAS_RB).
    // (128-row form with both a token row and a row constant: the row constants would push the ring into AGPR copies - an asm-load
    // destination must never be moved before its counted wait - so they are requested behind the K loops there)
    constexpr bool RB_LATE = RT >= 4 && HAS_TAB && HAS_RB;
    const int er = lane >> 2, eu = u0 + 4 * (lane & 3);
    f32x4 bi[3], bh[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bh[q] = *reinterpret_cast<const f32x4*>(a.b_hh + q * a.H + eu);
        bi[q] = a.b_ih ? *reinterpret_cast<const f32x4*>(a.b_ih + q * a.H + eu) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const long H3 = 3L * a.H;
    f32x4 hv[RT], tv[RT][3], rv[RT][3];
    int rows[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        rows[m] = m0 + 16 * m + er;
        const int rc = min(rows[m], a.B - 1);
        hv[m] = *reinterpret_cast<const f32x4*>(a.h_prev + (long)rc * a.ldh + eu);
        if (HAS_TAB) {
            const int tok = a.token(rc);
#pragma unroll
            for (int q = 0; q < 3; ++q) tv[m][q] = *reinterpret_cast<const f32x4*>(a.gx_table + (long)tok * H3 + q * a.H + eu);
        }
        if (HAS_RB && !RB_LATE) {
#pragma unroll
            for (int q = 0; q < 3; ++q) rv[m][q] = *reinterpret_cast<const f32x4*>(a.gx_rowbias + (long)rc * H3 + q * a.H + eu);
        }
    }
    if (a.x) phase(a.x, a.ldx, a.w_ih, a.ldw_ih, a.K1 >> 4, anx);
    phase(a.h_prev, a.ldh, a.w_hh, a.ldw_hh, a.H >> 4, anh);
    if (HAS_RB && RB_LATE) {
#pragma unroll
        for (int m = 0; m < RT; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) rv[m][q] = *reinterpret_cast<const f32x4*>(a.gx_rowbias + (long)min(rows[m], a.B - 1) * H3 + q * a.H + eu);
    }
    __shared__ __attribute__((aligned(16))) float tr[4][4][16 * 20];     // [wave][tile of the row tile: r, z, n(x), n(h)][16 rows x (16 + 4 pad)]
    float* tw = &tr[wave][0][0];
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        // D layout: lane (li, lg) holds rows 4 lg + i, column li
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tw[0 * 320 + (4 * lg + i) * 20 + li] = arz[m][0][i];
            tw[1 * 320 + (4 * lg + i) * 20 + li] = arz[m][1][i];
            tw[2 * 320 + (4 * lg + i) * 20 + li] = anx[m][i];
            tw[3 * 320 + (4 * lg + i) * 20 + li] = anh[m][i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f32x4 g_r = *reinterpret_cast<const f32x4*>(tw + 0 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_z = *reinterpret_cast<const f32x4*>(tw + 1 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_nx = *reinterpret_cast<const f32x4*>(tw + 2 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_nh = *reinterpret_cast<const f32x4*>(tw + 3 * 320 + er * 20 + 4 * (lane & 3));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float gi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float e = bi[q][c];
                if (HAS_TAB) e += tv[m][q][c];
                if (HAS_RB) e += rv[m][q][c];
                gi[q] = e;
            }
            const float r = fn_sigmoid((gi[0] + bh[0][c]) + g_r[c]);
            const float z = fn_sigmoid((gi[1] + bh[1][c]) + g_z[c]);
            const float n = fn_tanh((gi[2] + g_nx[c]) + r * (g_nh[c] + bh[2][c]));
            o[c] = (1.0f - z) * n + z * hv[m][c];
        }
        if (rows[m] < a.B) *reinterpret_cast<f32x4*>(a.h_out + (long)rows[m] * a.ldo + eu) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Output layer of the large-batch decode with the argmax in its epilogue (fn_out_argmax_f32): the same LDS-free loop; a wave owns
// 16 rows x 48 vocabulary columns (weight rows past V are the last row again, masked below), a workgroup 2 x 2 waves = 32 rows x 96
// columns - 256 workgroups at 2048 rows.  Lane (li, lg) ends up with the logits of rows 4 lg + i, columns n0 + 16 q + li: the best
// packed word of its three columns per row, a 16-lane butterfly over li, one 64-bit atomic max per row and wave.  The staged GEMM
// took 16 us for this 0.7-GFLOP product (16 barrier-separated tiles) and the separate argmax kernel 5 us per token.
FN_DEVINL unsigned long long fn_pack_best(float x, int v, int V) {
    const unsigned b = __float_as_uint(x);
    const unsigned key = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)key << 32) | (unsigned)(V - 1 - v);
}

template <int PF>
__global__ __launch_bounds__(NT, 2) void out_argmax_direct_kernel(const float* __restrict__ h, long ldh, const float* __restrict__ W, long ldw,
                                                                 const float* __restrict__ bias, int B, int V, int K,
                                                                 unsigned long long* __restrict__ best) {
    const int ncb = (V + 95) / 96;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x / ncb) * 32 + (wave >> 1) * 16, n0 = (blockIdx.x % ncb) * 96 + (wave & 1) * 48;
    const int li = lane & 15, lg = lane >> 4;
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned oa = (unsigned)(((long)min(m0 + li, B - 1) * ldh + 4 * lg) * 4);
    unsigned ob[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) ob[q] = (unsigned)(((long)min(n0 + 16 * q + li, V - 1) * ldw + 4 * lg) * 4);
    f32x4 fa[PF], fb[PF][3];
    const float* pa = h;
    const float* pb = W;
    auto load = [&](int set) {
        fn_gld4_sb(fa[set], oa, pa);
#pragma unroll
        for (int q = 0; q < 3; ++q) fn_gld4_sb(fb[set][q], ob[q], pb);
        pa += 16;
        pb += 16;
    };
    auto mma = [&](int u) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], fb[u][q][j], acc[q], 0, 0, 0);
    };
    auto keep = [&](int set) {
        fn_keep(fa[set]);
#pragma unroll
        for (int q = 0; q < 3; ++q) fn_keep(fb[set][q]);
    };
    const int nks = K >> 4, nmain = nks / PF * PF;
    if (nmain > 0) {
#pragma unroll
        for (int s = 0; s < PF; ++s) load(s);
        for (int base = 0; base + PF < nmain; base += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                fn_wait_vm<4 * (PF - 1)>();
                mma(u);
                load(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u == 0) fn_wait_vm<4 * (PF - 1)>();
            else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? 4 * (PF - 2) : 0)>();
            else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? 4 * (PF - 3) : 0)>();
            else fn_wait_vm<0>();
            mma(u);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < PF; ++s) keep(s);
    }
    for (int ks = nmain; ks < nks; ++ks) {
        load(0);
        fn_wait_vm<0>();
        mma(0);
        keep(0);
    }
    float bv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) bv[q] = bias[min(n0 + 16 * q + li, V - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long w = 0ull;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = n0 + 16 * q + li;
            const unsigned long long c = col < V ? fn_pack_best(acc[q][i] + bv[q], col, V) : 0ull;
            w = c > w ? c : w;
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)(w & 0xffffffffull), d, 64), hi = __shfl_xor((unsigned)(w >> 32), d, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            w = o > w ? o : w;
        }
        const int row = m0 + 4 * lg + i;
        if (li == 0 && row < B) atomicMax(best + row, w);
    }
}

// The same product with the WEIGHT operand resident in LDS: the LDS-free loops of this file all run at about 8 TB/s of operand loads
// (16 rows x 64 bytes per instruction), and at 16 x 48 outputs per wave that is 134 MB per launch - 17.6 us, no faster than the
// staged GEMM.  Here a workgroup owns 64 rows x 48 columns: its 48 weight rows (48 x K floats, 96 KB at K = 512) are laid out once as
// MFMA fragments [k step][column tile][lane] (16-byte LDS reads, conflict-free), only the state rows are read from memory in the loop
// (one 16-byte load per lane and 12 MFMAs, PF steps in flight).
template <int PF>
__global__ __launch_bounds__(NT) void out_argmax_lds_kernel(const float* __restrict__ h, long ldh, const float* __restrict__ W, long ldw,
                                                            const float* __restrict__ bias, int B, int V, int K,
                                                            unsigned long long* __restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x4* wl = reinterpret_cast<f32x4*>(dsm);       // [K / 16][3][64]
    const int ncb = (V + 47) / 48;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = (blockIdx.x / ncb) * 64 + wave * 16, n0 = (blockIdx.x % ncb) * 48;
    const int li = lane & 15, lg = lane >> 4;
    const int nks = K >> 4;
    const unsigned oa = (unsigned)(((long)min(m0 + li, B - 1) * ldh + 4 * lg) * 4);
    f32x4 fa[PF];
    const float* pa = h;
    auto load = [&](int set) {
        fn_gld4_sb(fa[set], oa, pa);
        pa += 16;
    };
    const int npro = min(PF, nks);
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < npro) load(s);
    // weight fragments: item (row r of 48, k quad c of K / 4) -> step c >> 2, tile r >> 4, lane group c & 3, slot (r & 15) rotated; consecutive threads read one row
    const int kq = K >> 2;
    // eight independent 16-byte loads per thread in flight, then their eight LDS writes (a load -> write pair per iteration serialised 24
    // memory latencies: 7.4 us of a 19 us launch); (row, quad) advance by NT items without a division
    const int total = 48 * kq, dq = NT / kq, dr = NT % kq;
    int fr = threadIdx.x / kq, fc = threadIdx.x % kq;
    for (int base = threadIdx.x; base < total; base += NT * 8) {
        f32x4 v[8];
        int rr[8], cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            rr[u] = fr;
            cc[u] = fc;
            v[u] = *reinterpret_cast<const f32x4*>(W + (long)min(n0 + min(fr, 47), V - 1) * ldw + 4 * fc);
            fc += dr;
            fr += dq;
            if (fc >= kq) { fc -= kq; ++fr; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rr[u], c = cc[u];
            if (r < 48) wl[(c >> 2) * 192 + (r >> 4) * 64 + (c & 3) * 16 + (((r & 15) + 4 * (c & 3) + (c >> 2)) & 15)] = v[u];     // slot rotated by the k quad: 16 consecutive quads of a row hit 16 bank groups
        }
    }
    __syncthreads();
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int u, int s) {
        const int sl = lg * 16 + ((li + 4 * lg + s) & 15);
        const f32x4 b0 = wl[s * 192 + sl], b1 = wl[s * 192 + 64 + sl], b2 = wl[s * 192 + 128 + sl];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b0[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b1[j], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b2[j], acc[2], 0, 0, 0);
        }
    };
    // the plain loads of the fill are older than nothing of the ring that is still needed: every counted wait below also covers them
    const int nmain = nks / PF * PF;
    int s = 0;
    if (nmain > 0) {
        for (; s + PF < nmain; s += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                fn_wait_vm<PF - 1>();
                mma(u, s + u);
                load(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the ring holds steps s .. s + PF - 1
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u == 0) fn_wait_vm<PF - 1>();
            else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? PF - 2 : 0)>();
            else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? PF - 3 : 0)>();
            else fn_wait_vm<0>();
            mma(u, s + u);
            __builtin_amdgcn_sched_barrier(0);
        }
        s += PF;
    } else {
        fn_wait_vm<0>();                             // fewer steps than ring slots (K < 16 PF)
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < npro) mma(u, u);
        s = npro;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) fn_keep(fa[u]);
    for (; s < nks; ++s) {                           // < PF leftover steps, unpipelined
        load(0);
        fn_wait_vm<0>();
        mma(0, s);
        fn_keep(fa[0]);
    }
    float bv[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) bv[q] = bias[min(n0 + 16 * q + li, V - 1)];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long w = 0ull;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = n0 + 16 * q + li;
            const unsigned long long c = col < V ? fn_pack_best(acc[q][i] + bv[q], col, V) : 0ull;
            w = c > w ? c : w;
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)(w & 0xffffffffull), d, 64), hi = __shfl_xor((unsigned)(w >> 32), d, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            w = o > w ? o : w;
        }
        const int row = m0 + 4 * lg + i;
        if (li == 0 && row < B) atomicMax(best + row, w);
    }
}

template <int PF>
__global__ __launch_bounds__(2 * NT) void out_argmax_lds8_kernel(const float* __restrict__ h, long ldh, const float* __restrict__ W, long ldw,
                                                            const float* __restrict__ bias, int B, int V, int K,
                                                            unsigned long long* __restrict__ best) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x4* wl = reinterpret_cast<f32x4*>(dsm);       // [K / 16][3][64]
    const int ncb = (V + 47) / 48;
    // eight waves: waves 4-7 take the second half of K for the row tiles of waves 0-3 and hand their accumulators over through LDS
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3, kh = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 8));
    const int m0 = (blockIdx.x / ncb) * 64 + wave * 16, n0 = (blockIdx.x % ncb) * 48;
    const int li = lane & 15, lg = lane >> 4;
    const int nks_all = K >> 4, s_first = kh ? nks_all / 2 : 0, nks = kh ? nks_all - nks_all / 2 : nks_all / 2;
    const unsigned oa = (unsigned)(((long)min(m0 + li, B - 1) * ldh + 4 * lg) * 4);
    f32x4 fa[PF];
    const float* pa = h + 16 * s_first;
    auto load = [&](int set) {
        fn_gld4_sb(fa[set], oa, pa);
        pa += 16;
    };
    float bv[3];                                     // requested in front of everything: the epilogue then waits for nothing
#pragma unroll
    for (int q = 0; q < 3; ++q) bv[q] = bias[min(n0 + 16 * q + li, V - 1)];
    const int npro = min(PF, nks);
#pragma unroll
    for (int s = 0; s < PF; ++s)
        if (s < npro) load(s);
    // weight fragments: item (row r of 48, k quad c of K / 4) -> step c >> 2, tile r >> 4, lane group c & 3, slot (r & 15) rotated; consecutive threads read one row
    const int kq = K >> 2;
    // eight independent 16-byte loads per thread in flight, then their eight LDS writes (a load -> write pair per iteration serialised 24
    // memory latencies: 7.4 us of a 19 us launch); (row, quad) advance by NT items without a division
    const int total = 48 * kq, dq = (2 * NT) / kq, dr = (2 * NT) % kq;
    int fr = threadIdx.x / kq, fc = threadIdx.x % kq;
    for (int base = threadIdx.x; base < total; base += 2 * NT * 8) {
        f32x4 v[8];
        int rr[8], cc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            rr[u] = fr;
            cc[u] = fc;
            v[u] = *reinterpret_cast<const f32x4*>(W + (long)min(n0 + min(fr, 47), V - 1) * ldw + 4 * fc);
            fc += dr;
            fr += dq;
            if (fc >= kq) { fc -= kq; ++fr; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = rr[u], c = cc[u];
            if (r < 48) wl[(c >> 2) * 192 + (r >> 4) * 64 + (c & 3) * 16 + (((r & 15) + 4 * (c & 3) + (c >> 2)) & 15)] = v[u];     // slot rotated by the k quad: 16 consecutive quads of a row hit 16 bank groups
        }
    }
    __syncthreads();
    f32x4 acc[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto mma = [&](int u, int srel) {
        const int s = s_first + srel;
        const int sl = lg * 16 + ((li + 4 * lg + s) & 15);
        const f32x4 b0 = wl[s * 192 + sl], b1 = wl[s * 192 + 64 + sl], b2 = wl[s * 192 + 128 + sl];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b0[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b1[j], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][j], b2[j], acc[2], 0, 0, 0);
        }
    };
    // the plain loads of the fill are older than nothing of the ring that is still needed: every counted wait below also covers them
    const int nmain = nks / PF * PF;
    int s = 0;
    if (nmain > 0) {
        for (; s + PF < nmain; s += PF) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                fn_wait_vm<PF - 1>();
                mma(u, s + u);
                load(u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the ring holds steps s .. s + PF - 1
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (u == 0) fn_wait_vm<PF - 1>();
            else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? PF - 2 : 0)>();
            else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? PF - 3 : 0)>();
            else fn_wait_vm<0>();
            mma(u, s + u);
            __builtin_amdgcn_sched_barrier(0);
        }
        s += PF;
    } else {
        fn_wait_vm<0>();                             // fewer steps than ring slots (K < 16 PF)
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (u < npro) mma(u, u);
        s = npro;
    }
#pragma unroll
    for (int u = 0; u < PF; ++u) fn_keep(fa[u]);
    for (; s < nks; ++s) {                           // < PF leftover steps, unpipelined
        load(0);
        fn_wait_vm<0>();
        mma(0, s);
        fn_keep(fa[0]);
    }
    {
        f32x4* rd = reinterpret_cast<f32x4*>(dsm + (long)48 * K) + (wave * 3) * 64 + lane;     // behind the weight fragments
        if (kh) {
#pragma unroll
            for (int q = 0; q < 3; ++q) rd[q * 64] = acc[q];
        }
        __syncthreads();
        if (kh) return;
#pragma unroll
        for (int q = 0; q < 3; ++q) acc[q] += rd[q * 64];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned long long w = 0ull;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int col = n0 + 16 * q + li;
            const unsigned long long c = col < V ? fn_pack_best(acc[q][i] + bv[q], col, V) : 0ull;
            w = c > w ? c : w;
        }
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const unsigned lo = __shfl_xor((unsigned)(w & 0xffffffffull), d, 64), hi = __shfl_xor((unsigned)(w >> 32), d, 64);
            const unsigned long long o = ((unsigned long long)hi << 32) | lo;
            w = o > w ? o : w;
        }
        const int row = m0 + 4 * lg + i;
        if (li == 0 && row < B) atomicMax(best + row, w);
    }
}

__global__ __launch_bounds__(256) void best_tokens_kernel(const unsigned long long* __restrict__ best, long n, int B, int V, int* __restrict__ tokens, long tok_ld) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const long t = i / B, b = i - t * B;
        tokens[b * tok_ld + t] = V - 1 - (int)(unsigned)(best[i] & 0xffffffffull);
    }
}

static bool cell_direct_ok(const CellArgs& a) {
    auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if ((a.H % 32) != 0 || !al(a.h_prev) || !al(a.w_hh) || (a.ldh & 3) || (a.ldw_hh & 3)) return false;
    if (!al(a.h_out) || (a.ldo & 3) || !al(a.b_hh) || (a.b_ih && !al(a.b_ih)) || (a.gx_table && !al(a.gx_table)) || (a.gx_rowbias && !al(a.gx_rowbias))) return false;
    if ((long)a.B * a.ldh * 4 >= (1L << 32) || 3L * a.H * a.ldw_hh * 4 >= (1L << 32)) return false;
    if (a.x) {
        if ((a.K1 % 16) != 0 || !al(a.x) || !al(a.w_ih) || (a.ldx & 3) || (a.ldw_ih & 3)) return false;
        if ((long)a.B * a.ldx * 4 >= (1L << 32) || 3L * a.H * a.ldw_ih * 4 >= (1L << 32)) return false;
    }
    return true;
}

template <int RT, int PF>
int launch_cell_direct(const CellArgs& a, hipStream_t st) {
    const int tiles = ((a.B + 32 * RT - 1) / (32 * RT)) * (a.H / 32);
    const bool tab = a.gx_table != nullptr, rb = a.gx_rowbias != nullptr;
    if (tab && rb) hipLaunchKernelGGL((gru_cell_direct_kernel<RT, PF, true, true>), dim3(tiles), dim3(NT), 0, st, a);
    else if (tab) hipLaunchKernelGGL((gru_cell_direct_kernel<RT, PF, true, false>), dim3(tiles), dim3(NT), 0, st, a);
    else if (rb) hipLaunchKernelGGL((gru_cell_direct_kernel<RT, PF, false, true>), dim3(tiles), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((gru_cell_direct_kernel<RT, PF, false, false>), dim3(tiles), dim3(NT), 0, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

// Gate epilogue of the cells with an LDS-resident weight slice, on (row, 4 units) items: the accumulator tiles of row tile m go through the
// wave-private LDS tile tw (D layout in, lane (row = l >> 2, units 4 (l & 3) ..) out), every other operand - old state, token row, row
// constant - is ONE 16-byte load per gate and item, the new state one 16-byte store.  tok[m]: the token of the lane's row of tile m.
// biases and old-state rows of the epilogue items: requested in front of the last K phase where the registers allow it (EARLY)
template <int RT>
struct CellEpiPre { f32x4 bi[3], bh[3], hv[RT]; };
template <int RT>
FN_DEVINL void cell_epilogue_request(const CellArgs& a, int m0, int u0, int lane, CellEpiPre<RT>& p) {
    const int er = lane >> 2, eu = u0 + 4 * (lane & 3);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        p.bh[q] = *reinterpret_cast<const f32x4*>(a.b_hh + q * a.H + eu);
        p.bi[q] = a.b_ih ? *reinterpret_cast<const f32x4*>(a.b_ih + q * a.H + eu) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int m = 0; m < RT; ++m) p.hv[m] = *reinterpret_cast<const f32x4*>(a.h_prev + (long)min(m0 + 16 * m + er, a.B - 1) * a.ldh + eu);
}

template <int RT, bool HAS_TAB, bool HAS_RB>
FN_DEVINL void cell_epilogue_items(const CellArgs& a, int m0, int u0, int lane, float* tw, const int (&tok)[RT], const f32x4 (&arz)[RT][2],
                                   const f32x4 (&anx)[RT], const f32x4 (&anh)[RT], const CellEpiPre<RT>& pre) {
    const int li = lane & 15, lg = lane >> 4;
    const int er = lane >> 2, eu = u0 + 4 * (lane & 3);
    const f32x4 (&bi)[3] = pre.bi;
    const f32x4 (&bh)[3] = pre.bh;
    const f32x4 (&hv)[RT] = pre.hv;
    const long H3 = 3L * a.H;
    f32x4 tv[RT][3], rv[RT][3];
    int rows[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) {
        rows[m] = m0 + 16 * m + er;
        const int rc = min(rows[m], a.B - 1);
        if (HAS_TAB) {
#pragma unroll
            for (int q = 0; q < 3; ++q) tv[m][q] = *reinterpret_cast<const f32x4*>(a.gx_table + (long)tok[m] * H3 + q * a.H + eu);
        }
        if (HAS_RB) {
#pragma unroll
            for (int q = 0; q < 3; ++q) rv[m][q] = *reinterpret_cast<const f32x4*>(a.gx_rowbias + (long)rc * H3 + q * a.H + eu);
        }
    }
#pragma unroll
    for (int m = 0; m < RT; ++m) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            tw[0 * 320 + (4 * lg + i) * 20 + li] = arz[m][0][i];
            tw[1 * 320 + (4 * lg + i) * 20 + li] = arz[m][1][i];
            tw[2 * 320 + (4 * lg + i) * 20 + li] = anx[m][i];
            tw[3 * 320 + (4 * lg + i) * 20 + li] = anh[m][i];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const f32x4 g_r = *reinterpret_cast<const f32x4*>(tw + 0 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_z = *reinterpret_cast<const f32x4*>(tw + 1 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_nx = *reinterpret_cast<const f32x4*>(tw + 2 * 320 + er * 20 + 4 * (lane & 3));
        const f32x4 g_nh = *reinterpret_cast<const f32x4*>(tw + 3 * 320 + er * 20 + 4 * (lane & 3));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        f32x4 o;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float gi[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float e = bi[q][c];
                if (HAS_TAB) e += tv[m][q][c];
                if (HAS_RB) e += rv[m][q][c];
                gi[q] = e;
            }
            const float r = fn_sigmoid((gi[0] + bh[0][c]) + g_r[c]);
            const float z = fn_sigmoid((gi[1] + bh[1][c]) + g_z[c]);
            const float n = fn_tanh((gi[2] + g_nx[c]) + r * (g_nh[c] + bh[2][c]));
            o[c] = (1.0f - z) * n + z * hv[m][c];
        }
        if (rows[m] < a.B) *reinterpret_cast<f32x4*>(a.h_out + (long)rows[m] * a.ldo + eu) = o;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The cell with the WEIGHT operand of a K phase resident in LDS (as out_argmax_lds_kernel): the LDS-free loop above is bound by its
// operand loads (~8 TB/s of 16 x 64-byte pieces per instruction: 7 loads per 48 MFMAs), not by the MFMA pipe.  Here a workgroup owns
// 64 RT rows x 16 hidden units: the 48 weight rows of its units (48 x K floats, 96 KB at K = 512) are laid out once per phase as MFMA
// fragments [k step][gate][lane] (16-byte LDS reads, slots rotated against bank conflicts), and only the RT state-row tiles of a wave are
// read from memory in the loop (RT loads per 12 RT MFMAs).  Layer 2 fills the slice twice (W_ih, then W_hh after a barrier).
template <int RT, int PF, bool HAS_TAB, bool HAS_RB>
__global__ __launch_bounds__(NT) void gru_cell_wlds_kernel(const CellArgs a, int kmax) {
    static_assert(PF >= 2 && (PF % 2) == 0, "the fragment double buffer follows the parity of the ring slot");
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x4* wl = reinterpret_cast<f32x4*>(dsm);       // [K / 16][3][64]
    float* tr = dsm + (long)48 * kmax;               // [wave][4 tiles][16 x 20]
    const int nut = a.H >> 4;
    const int tm = blockIdx.x / nut, tu = blockIdx.x % nut;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = tm * 64 * RT + wave * 16 * RT, u0 = tu * 16;
    const int li = lane & 15, lg = lane >> 4;
    f32x4 arz[RT][2], anx[RT], anh[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) arz[m][0] = arz[m][1] = anx[m] = anh[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 fa[PF][RT];
    auto phase = [&](const float* A, long lda, const float* W, long ldw, int K, f32x4 (&accn)[RT], bool refill) {
        const int nks = K >> 4, kq = K >> 2;
        unsigned oa[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) oa[m] = (unsigned)(((long)min(m0 + 16 * m + li, a.B - 1) * lda + 4 * lg) * 4);
        const float* pa = A;
        auto load = [&](int set) {
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_gld4_sb(fa[set][m], oa[m], pa);
            pa += 16;
        };
        const int npro = min(PF, nks);
#pragma unroll
        for (int s = 0; s < PF; ++s)
            if (s < npro) load(s);
        if (refill) __syncthreads();                 // every wave has left the previous phase's loop
        // weight fragments: item (row r of 48 = gate r >> 4, unit r & 15; k quad c) -> step c >> 2, gate, lane group c & 3, slot (r & 15) rotated
        const int total = 48 * kq, dq = NT / kq, dr = NT % kq;
        int fr = threadIdx.x / kq, fc = threadIdx.x % kq;
        for (int base = threadIdx.x; base < total; base += NT * 8) {
            f32x4 v[8];
            int rr[8], cc[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                rr[u] = fr;
                cc[u] = fc;
                const int r = min(fr, 47);
                v[u] = *reinterpret_cast<const f32x4*>(W + (long)((r >> 4) * a.H + u0 + (r & 15)) * ldw + 4 * fc);
                fc += dr;
                fr += dq;
                if (fc >= kq) { fc -= kq; ++fr; }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = rr[u], c = cc[u];
                if (r < 48) wl[(c >> 2) * 192 + (r >> 4) * 64 + (c & 3) * 16 + (((r & 15) + 4 * (c & 3) + (c >> 2)) & 15)] = v[u];
            }
        }
        __syncthreads();
        // weight fragments of step s + 1 are read while the MFMAs of step s run (bq[parity]); PF is even, so the parity of a step is its ring slot's
        f32x4 bq[2][3];
        auto bread = [&](int buf, int s) {
            const int sl = lg * 16 + ((li + 4 * lg + s) & 15);
            bq[buf][0] = wl[s * 192 + sl];
            bq[buf][1] = wl[s * 192 + 64 + sl];
            bq[buf][2] = wl[s * 192 + 128 + sl];
        };
        auto mma = [&](int u, int s) {
            if (s + 1 < nks) bread((u & 1) ^ 1, s + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    arz[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], bq[u & 1][0][j], arz[m][0], 0, 0, 0);
                    arz[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], bq[u & 1][1][j], arz[m][1], 0, 0, 0);
                    accn[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u][m][j], bq[u & 1][2][j], accn[m], 0, 0, 0);
                }
        };
        bread(0, 0);
        const int nmain = nks / PF * PF;
        int s = 0;
        if (nmain > 0) {
            for (; s + PF < nmain; s += PF) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    fn_wait_vm<RT * (PF - 1)>();
                    mma(u, s + u);
                    load(u);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                if (u == 0) fn_wait_vm<RT * (PF - 1)>();
                else if (u == 1 && PF > 1) fn_wait_vm<(PF > 1 ? RT * (PF - 2) : 0)>();
                else if (u == 2 && PF > 2) fn_wait_vm<(PF > 2 ? RT * (PF - 3) : 0)>();
                else fn_wait_vm<0>();
                mma(u, s + u);
                __builtin_amdgcn_sched_barrier(0);
            }
            s += PF;
        } else {
            fn_wait_vm<0>();                         // fewer steps than ring slots
#pragma unroll
            for (int u = 0; u < PF; ++u)
                if (u < npro) mma(u, u);
            s = npro;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_keep(fa[u][m]);
        for (; s < nks; ++s) {                       // < PF leftover steps, unpipelined; ring slot = parity of the step (bq[parity] holds its fragments)
            if (s & 1) {
                load(1);
                fn_wait_vm<0>();
                mma(1, s);
            } else {
                load(0);
                fn_wait_vm<0>();
                mma(0, s);
            }
#pragma unroll
            for (int m = 0; m < RT; ++m) { fn_keep(fa[0][m]); fn_keep(fa[1][m]); }
        }
    };
    // the token of every epilogue row now (the table-row loads of the epilogue then cost one memory latency, not two)
    int tok[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) tok[m] = HAS_TAB ? a.token(min(m0 + 16 * m + (lane >> 2), a.B - 1)) : 0;
    if (a.x) phase(a.x, a.ldx, a.w_ih, a.ldw_ih, a.K1, anx, false);
    phase(a.h_prev, a.ldh, a.w_hh, a.ldw_hh, a.H, anh, a.x != nullptr);
    CellEpiPre<RT> pre;
    cell_epilogue_request<RT>(a, m0, u0, lane, pre);
    cell_epilogue_items<RT, HAS_TAB, HAS_RB>(a, m0, u0, lane, tr + wave * 4 * 320, tok, arz, anx, anh, pre);
}

template <int RT, int PF, bool HAS_TAB, bool HAS_RB>
int launch_cell_wlds_inst(const CellArgs& a, int kmax, size_t lds, hipStream_t st) {
    static std::atomic<bool> attr_set[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    auto k = gru_cell_wlds_kernel<RT, PF, HAS_TAB, HAS_RB>;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int tiles = ((a.B + 64 * RT - 1) / (64 * RT)) * (a.H / 16);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), lds, st, a, kmax);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

static bool cell_wlds_ok(const CellArgs& a) {
    const int kmax = a.x ? (a.K1 > a.H ? a.K1 : a.H) : a.H;
    return (size_t)48 * kmax * 4 + 4 * 4 * 320 * 4 <= (size_t)160 * 1024;
}

template <int RT, int PF>
int launch_cell_wlds(const CellArgs& a, hipStream_t st) {
    const int kmax = a.x ? (a.K1 > a.H ? a.K1 : a.H) : a.H;
    const size_t lds = (size_t)48 * kmax * 4 + 4 * 4 * 320 * 4;
    const bool tab = a.gx_table != nullptr, rb = a.gx_rowbias != nullptr;
    if (tab && rb) return launch_cell_wlds_inst<RT, PF, true, true>(a, kmax, lds, st);
    if (tab) return launch_cell_wlds_inst<RT, PF, true, false>(a, kmax, lds, st);
    if (rb) return launch_cell_wlds_inst<RT, PF, false, true>(a, kmax, lds, st);
    return launch_cell_wlds_inst<RT, PF, false, false>(a, kmax, lds, st);
}

// ---------------------------------------------------------------------------------------------------------------
// One GRUCell step of a large batch on the bf16 MFMA with exact triple splits (round 6; FnGruCell.variant bit 14): the producer / consumer form of the
// round-6 GEMMs (x6w_core.h, gemm.hip) with the gate epilogue behind it.  Workgroup = 128 rows x 32 hidden units; its "N tile" is 128 columns:
// [r | z | n_x | n_h] of those 32 units, where r and z accumulate the dense-input part x W_ih^T AND the recurrent part h W_hh^T (one K loop over
// K1 + H), n_x only the input part and n_h only the recurrent part (their weight rows count as zeros in the other part: a quarter of the MFMAs multiply
// zeros - the cell is bound by its prologue / epilogue and the loads, not by the matrix pipe, so the uniform loop wins).  Waves 4-7 load + split
// (A: rows of x then of h_prev, truncated pieces; B: weight rows, rounded pieces) into LDS, waves 0-3 multiply; the four 64 x 64 accumulator
// tiles then meet in LDS (the stages are free by then) and every thread finishes four (row, 4 units) items: biases, token row, row bias, gates,
// new state.  B % 128 == 0, H % 32 == 0, K1 % 32 == 0, 16-byte aligned operands.  Same gate arithmetic as the fp32 cells; the products are summed
// in another order (fp32-class): greedy tokens agree up to the reference's own near-ties (tests/test_gpu_parity.py).
// ---------------------------------------------------------------------------------------------------------------
template <bool RN>
FN_DEVINL void cell_x6_produce(u32x4* __restrict__ lds, const float* const (&px)[4], const float* const (&ph)[4], bool zero_x, bool zero_h, int nbx, int ps, int lane,
                               int nblk) {
    constexpr int NS = X6W_NS;
    const int r = lane >> 2, q = lane & 3;
    f32x4 fa[NS][8];                                     // [set][2 e + j]: set row 4 r + e, k = 8 q + 4 j ..
    auto gload = [&](auto SET, int blk) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const bool xp = blk < nbx;
        const long off = xp ? 32L * blk : 32L * (blk - nbx);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float* p0 = (xp ? px[e] : ph[e]) + off;
            fa[set][2 * e] = x6w_ld(p0);
            fa[set][2 * e + 1] = x6w_ld(p0 + 4);
        }
    };
    auto cut = [&](auto SET, int blk, int stage) __attribute__((always_inline)) {
        constexpr int set = decltype(SET)::value;
        const bool zero = blk < nbx ? zero_x : zero_h;   // this lane's rows do not take part in this K part: zeros (the loads above fetched legal rows)
        u32x4* dst = lds + stage * X6W_STAGE + ps * X6W_SET + r + 16 * q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x[8], hi[8], r1[8], mi[8], r2[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = zero ? 0.f : fa[set][2 * e + (j >> 2)][j & 3];
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
    // the trip structure of x6w_produce_nt (gemm.hip): blocks 0, 1 cut and 2 .. NS requested, then per trip request t + NS + 1 | cut t + 2 | barrier
    x6w_for<NS>([&](auto I) __attribute__((always_inline)) {
        if (decltype(I)::value < nblk) gload(I, decltype(I)::value);
    });
    cut(x6w_ic<0>{}, 0, 0);
    if (nblk > 1) cut(x6w_ic<1>{}, 1, 1);
    if (nblk > NS) gload(x6w_ic<0>{}, NS);
    x6w_barrier_p();
    int stage = 2, t = 0;
#pragma unroll 1
    for (; t + 2 * NS < nblk; t += NS) {
        x6w_for<NS>([&](auto R) __attribute__((always_inline)) {
            constexpr int rr = decltype(R)::value;
            gload(x6w_ic<(rr + 1) % NS>{}, t + rr + NS + 1);
            cut(x6w_ic<(rr + 2) % NS>{}, t + rr + 2, stage);
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        });
    }
    x6w_for<2 * NS>([&](auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value, rr = i % NS;
        if (t + i < nblk) {
            if (t + i + NS + 1 < nblk) gload(x6w_ic<(rr + 1) % NS>{}, t + i + NS + 1);
            if (t + i + 2 < nblk) cut(x6w_ic<(rr + 2) % NS>{}, t + i + 2, stage);
            stage = stage == 2 ? 0 : stage + 1;
            x6w_barrier_p();
        }
    });
}

constexpr int CX_LDT = 68;                           // floats per row of an accumulator tile in LDS (64 + 4: 16-byte rows, conflict-free column reads)
template <bool HAS_TAB, bool HAS_RB>
__global__ __launch_bounds__(X6W_NT) void gru_cell_x6_kernel(const CellArgs a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 cx_lds[];       // [3 stages][4 sets][4 tiles][3 pieces][64 lanes]; epilogue: float [2][2][64][CX_LDT]
    const int nut = a.H >> 5;
    const int v = fn_xcd_remap(blockIdx.x, gridDim.x);                   // consecutive unit tiles of one row panel on one XCD (they share the rows of x / h_prev)
    const int m0 = (v / nut) * 128, u0 = (v % nut) * 32;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nbx = a.x ? a.K1 >> 5 : 0, nblk = nbx + (a.H >> 5);
    if (wave >= 4) {
        const int ps = wave & 3, r = lane >> 2, q = lane & 3;
        const float* px[4];
        const float* ph[4];
        bool zx = false, zh = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int s_ = 4 * r + e;                                    // row of the 64-row set
            if (ps < 2) {                                                // A: batch rows m0 + 64 ps + s_
                const long row = m0 + 64 * ps + s_;
                px[e] = (a.x ? a.x + row * a.ldx : a.h_prev + row * a.ldh) + 8 * q;
                ph[e] = a.h_prev + row * a.ldh + 8 * q;
            } else {                                                     // B: set 2 = [r | z], set 3 = [n_x | n_h] of units u0 .. u0 + 31
                const int half = s_ >> 5, u = u0 + (s_ & 31);
                const int gate = ps == 2 ? half : 2;
                px[e] = (a.x ? a.w_ih + (long)(gate * a.H + u) * a.ldw_ih : a.w_hh + (long)(gate * a.H + u) * a.ldw_hh) + 8 * q;
                ph[e] = a.w_hh + (long)(gate * a.H + u) * a.ldw_hh + 8 * q;
                if (ps == 3) { zx = half == 1; zh = half == 0; }        // n_x takes the input part only, n_h the recurrent part only
            }
        }
        if (ps < 2) cell_x6_produce<false>(cx_lds, px, ph, zx, zh, nbx, ps, lane, nblk);
        else cell_x6_produce<true>(cx_lds, px, ph, zx, zh, nbx, ps, lane, nblk);
        // ---- the producers are done two blocks before the consumers: they fetch the epilogue's operands meanwhile and finish the cell.
        //      Thread = four (row, 4 units) items: rows m0 + tid / 8 + 32 k, units u0 + 4 (tid % 8) ----
        const int tid = threadIdx.x - 256;
        const int uq = tid & 7, eu = u0 + 4 * uq;
        const long H3 = 3L * a.H;
        f32x4 bi[3], bh[3], hv[4], tv[4][3], rv[4][3];
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            bh[g] = *reinterpret_cast<const f32x4*>(a.b_hh + g * a.H + eu);
            bi[g] = a.b_ih ? *reinterpret_cast<const f32x4*>(a.b_ih + g * a.H + eu) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = m0 + (tid >> 3) + 32 * k;
            hv[k] = *reinterpret_cast<const f32x4*>(a.h_prev + (long)row * a.ldh + eu);
            if (HAS_TAB) {
                const int tok = a.token(row);
#pragma unroll
                for (int g = 0; g < 3; ++g) tv[k][g] = *reinterpret_cast<const f32x4*>(a.gx_table + (long)tok * H3 + g * a.H + eu);
            }
            if (HAS_RB) {
#pragma unroll
                for (int g = 0; g < 3; ++g) rv[k][g] = *reinterpret_cast<const f32x4*>(a.gx_rowbias + (long)row * H3 + g * a.H + eu);
            }
        }
        x6w_barrier_p();                                 // E0: every consumer has read its last operands - the stages may be overwritten
        x6w_barrier_p();                                 // E1: the four accumulator tiles are in LDS
        const float* T = reinterpret_cast<const float*>(cx_lds);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int lr = (tid >> 3) + 32 * k;          // row inside the 128-row panel
            const float* t0 = T + ((lr >> 6) * 2 + 0) * 64 * CX_LDT + (lr & 63) * CX_LDT;      // [r | z] tile of this row's panel half
            const float* t1 = t0 + 64 * CX_LDT;                                                   // [n_x | n_h]
            const f32x4 g_r = *reinterpret_cast<const f32x4*>(t0 + 4 * uq), g_z = *reinterpret_cast<const f32x4*>(t0 + 32 + 4 * uq);
            const f32x4 g_nx = *reinterpret_cast<const f32x4*>(t1 + 4 * uq), g_nh = *reinterpret_cast<const f32x4*>(t1 + 32 + 4 * uq);
            f32x4 o;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float gi[3];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    float e = bi[g][c];
                    if (HAS_TAB) e += tv[k][g][c];
                    if (HAS_RB) e += rv[k][g][c];
                    gi[g] = e;
                }
                const float rg = fn_sigmoid((gi[0] + bh[0][c]) + g_r[c]);
                const float zg = fn_sigmoid((gi[1] + bh[1][c]) + g_z[c]);
                const float ng = fn_tanh((gi[2] + g_nx[c]) + rg * (g_nh[c] + bh[2][c]));
                o[c] = (1.0f - zg) * ng + zg * hv[k][c];
            }
            *reinterpret_cast<f32x4*>(a.h_out + (long)(m0 + lr) * a.ldo + eu) = o;
        }
        return;
    }
    const int wm = (wave >> 1) & 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[i][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
    x6w_consume(cx_lds, wm, wn, lane, nblk, acc);
    x6w_barrier_p();                                     // E0
    {
        float* tw = reinterpret_cast<float*>(cx_lds) + (wm * 2 + wn) * 64 * CX_LDT;
        const int li = lane & 15, lg = lane >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const f32x4 o = {acc[i][0][rr], acc[i][1][rr], acc[i][2][rr], acc[i][3][rr]};
                *reinterpret_cast<f32x4*>(tw + (4 * (lg * 4 + rr) + i) * CX_LDT + 4 * li) = o;      // row 4 (lg 4 + rr) + i, columns 4 li .. of the wave's 64
            }
    }
    x6w_barrier_p();                                     // E1: the producers take it from here
}

static bool cell_x6_ok(const CellArgs& a) {
    const uintptr_t al = (uintptr_t)a.x | (uintptr_t)a.w_ih | (uintptr_t)a.h_prev | (uintptr_t)a.w_hh | (uintptr_t)a.h_out | (uintptr_t)a.b_hh | (uintptr_t)a.b_ih |
                         (uintptr_t)a.gx_table | (uintptr_t)a.gx_rowbias;
    return (a.B % 128) == 0 && (a.H % 32) == 0 && a.H >= 64 && (!a.x || ((a.K1 % 32) == 0 && (a.ldx & 3) == 0 && (a.ldw_ih & 3) == 0)) && (a.ldh & 3) == 0 &&
           (a.ldw_hh & 3) == 0 && (a.ldo & 3) == 0 && (al & 15) == 0;
}

template <bool HAS_TAB, bool HAS_RB>
int launch_cell_x6_inst(const CellArgs& a, hipStream_t st) {
    const size_t lds = (size_t)X6W_STAGES * X6W_STAGE * 16;
    static std::atomic<bool> attr_set[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    auto k = gru_cell_x6_kernel<HAS_TAB, HAS_RB>;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    hipLaunchKernelGGL(k, dim3((a.B / 128) * (a.H / 32)), dim3(X6W_NT), lds, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int launch_cell_x6(const CellArgs& a, hipStream_t st) {
    const bool tab = a.gx_table != nullptr, rb = a.gx_rowbias != nullptr;
    if (tab && rb) return launch_cell_x6_inst<true, true>(a, st);
    if (tab) return launch_cell_x6_inst<true, false>(a, st);
    if (rb) return launch_cell_x6_inst<false, true>(a, st);
    return launch_cell_x6_inst<false, false>(a, st);
}

FN_DEVINL void fn_wait_vm_n(int n) {                // n folds to a constant in fully unrolled loops
    switch (n) {
#define FN_WV(k) case k: fn_wait_vm<k>(); break;
        FN_WV(0) FN_WV(1) FN_WV(2) FN_WV(3) FN_WV(4) FN_WV(5) FN_WV(6) FN_WV(7) FN_WV(8) FN_WV(9) FN_WV(10) FN_WV(11) FN_WV(12)
        FN_WV(13) FN_WV(14) FN_WV(15) FN_WV(16) FN_WV(17) FN_WV(18) FN_WV(19) FN_WV(20)
#undef FN_WV
        default: fn_wait_vm<0>(); break;
    }
}

// gru_cell_wlds_kernel for K1 = H = 512 with the slice fills UNDER the K loops: the slice is filled in halves of 16 k steps; only the first
// half of the first phase is filled in front of its loop, every other half is requested one 16-byte load per thread and step during the
// 16 steps before it is needed (asm loads into 12 registers, counted together with the state-row ring), written to LDS and published by one
// barrier at the half boundary - layer 2 pays 1.5 us of fill instead of 6.  Fully unrolled (the counted waits differ from step to step).
template <int RT, int PF, bool HAS_TAB, bool HAS_RB>
__global__ __launch_bounds__(NT) void gru_cell_wlds_ovl_kernel(const CellArgs a) {
    constexpr int K = 512, NKS = K / 16, HALF = NKS / 2, NF = 12;     // NF = 48 rows x 64 quads of a half / 256 threads
    static_assert(PF >= 2 && (PF % 2) == 0 && PF <= 4 && NF + PF <= HALF, "ring / fill schedule");
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x4* wl = reinterpret_cast<f32x4*>(dsm);       // [K / 16][3][64]
    float* tr = dsm + 48 * K;
    const int nut = a.H >> 4;
    const int tm = blockIdx.x / nut, tu = blockIdx.x % nut;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m0 = tm * 64 * RT + wave * 16 * RT, u0 = tu * 16;
    const int li = lane & 15, lg = lane >> 4;
    f32x4 arz[RT][2], anx[RT], anh[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) arz[m][0] = arz[m][1] = anx[m] = anh[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 fa[PF][RT], fv[NF];
    // fill item j of a thread: weight row r = wave + 4 j of the 48 (gate r >> 4 = j >> 2, unit r & 15 = wave + 4 (j & 3)), k quad 64 half + lane:
    // one lane offset for all items, the item's part of the address in a scalar base (no vector address arithmetic, no pointer registers)
    auto f_off = [&](long ldw) { return (unsigned)((((long)(u0 + wave) * ldw) + 4 * lane) * 4); };
    auto f_base = [&](const float* W, long ldw, int half, int j) { return W + ((long)(j >> 2) * a.H + 4 * (j & 3)) * ldw + 256 * half; };
    auto fill_store = [&](int half) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int r = wave + 4 * j, c = 64 * half + lane;
            wl[(c >> 2) * 192 + (r >> 4) * 64 + (c & 3) * 16 + (((r & 15) + 4 * (c & 3) + (c >> 2)) & 15)] = fv[j];
        }
    };
    auto phase = [&](const float* A, long lda, const float* W, long ldw, f32x4 (&accn)[RT], auto FIRST, auto PRE2, const float* Wn, long ldwn) {
        constexpr bool first = decltype(FIRST)::value, pre2 = decltype(PRE2)::value;
        unsigned oa[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) oa[m] = (unsigned)(((long)min(m0 + 16 * m + li, a.B - 1) * lda + 4 * lg) * 4);
        const float* pa = A;
        auto load = [&](int set) {
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_gld4_sb(fa[set][m], oa[m], pa);
            pa += 16;
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) load(s);
        if (first) {
#pragma unroll
            for (int j = 0; j < NF; ++j) fv[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(f_base(W, ldw, 0, j)) + f_off(ldw));
        }
        // Not the first phase: this half was requested by asm loads during the previous phase's last steps, and its LAST load was issued behind the ring
        // loads of that phase's final steps - no ring wait covered it.  Everything but the PF ring requests just issued has to have landed before the
        // registers go to LDS (round 5: found by a soak - 1 launch in ~30 suites wrote a stale register: a wrong weight row for one (gate, unit)).
        if (!first) fn_wait_vm_n(PF * RT);
        fill_store(0);                               // (every reader of these steps' fragments passed the previous phase's half barrier)
        __syncthreads();
        f32x4 bq[2][3];
        auto bread = [&](int buf, int s) {
            const int sl = lg * 16 + ((li + 4 * lg + s) & 15);
            bq[buf][0] = wl[s * 192 + sl];
            bq[buf][1] = wl[s * 192 + 64 + sl];
            bq[buf][2] = wl[s * 192 + 128 + sl];
        };
        bread(0, 0);
        const unsigned fo = f_off(ldw), fon = pre2 ? f_off(ldwn) : 0u;
        auto step = [&](const int u) __attribute__((always_inline)) {
            // loads younger than the ring loads of step u: the fill load issued with them, then everything issued behind steps u - PF + 1 .. u - 1
            auto fillf = [&](int j) { return j < HALF ? (j < NF ? 1 : 0) : ((pre2 && j - HALF < NF) ? 1 : 0); };
            auto ringf = [&](int j) { return j + PF < NKS ? RT : 0; };
            int allowed = u >= PF ? fillf(u - PF) : (PF - 1 - u) * RT;
#pragma unroll
            for (int d = 1; d < PF; ++d)
                if (u - d >= 0) allowed += ringf(u - d) + fillf(u - d);
            fn_wait_vm_n(allowed);
            if (u != HALF - 1 && u != NKS - 1) bread((u & 1) ^ 1, u + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < RT; ++m) {
                    arz[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][0][j], arz[m][0], 0, 0, 0);
                    arz[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][1][j], arz[m][1], 0, 0, 0);
                    accn[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][2][j], accn[m], 0, 0, 0);
                }
            if (u + PF < NKS) load(u % PF);
            if (u < HALF) {
                if (u < NF) fn_gld4_sb(fv[u], fo, f_base(W, ldw, 1, u));
            } else if (pre2 && u - HALF < NF) {
                fn_gld4_sb(fv[u - HALF], fon, f_base(Wn, ldwn, 0, u - HALF));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int u = 0; u < HALF; ++u) step(u);
        // second half of the slice ...
        // ... requested during steps 0 .. NF - 1; the last of them was issued BEHIND the ring loads of step NF - 1 (for step NF - 1 + PF), so the ring
        // waits of the steps up to HALF - 1 do not cover it: wait until only the ring requests issued behind it (steps NF .. HALF - 1) are in flight
        fn_wait_vm_n((HALF - NF) * RT);
#pragma unroll
        for (int j = 0; j < NF; ++j) fn_keep(fv[j]);
        fill_store(1);
        __syncthreads();
        bread(0, HALF);
#pragma unroll
        for (int u = HALF; u < NKS; ++u) step(u);
#pragma unroll
        for (int s = 0; s < PF; ++s)
#pragma unroll
            for (int m = 0; m < RT; ++m) fn_keep(fa[s][m]);
        if (pre2) {
#pragma unroll
            for (int j = 0; j < NF; ++j) fn_keep(fv[j]);
        }
    };
    // the token of every epilogue row now (the table-row loads of the epilogue then cost one memory latency, not two)
    int tok[RT];
#pragma unroll
    for (int m = 0; m < RT; ++m) tok[m] = HAS_TAB ? a.token(min(m0 + 16 * m + (lane >> 2), a.B - 1)) : 0;
    using T_ = std::integral_constant<bool, true>;
    using F_ = std::integral_constant<bool, false>;
    constexpr bool EARLY = RT <= 2;                  // (the 192- and 256-row forms have no registers left beside their rings: no AGPR copies next to asm loads)
    CellEpiPre<RT> pre;
    if (a.x) {
        phase(a.x, a.ldx, a.w_ih, a.ldw_ih, anx, T_{}, T_{}, a.w_hh, a.ldw_hh);
        if (EARLY) cell_epilogue_request<RT>(a, m0, u0, lane, pre);
        phase(a.h_prev, a.ldh, a.w_hh, a.ldw_hh, anh, F_{}, F_{}, nullptr, 0);
    } else {
        if (EARLY) cell_epilogue_request<RT>(a, m0, u0, lane, pre);
        phase(a.h_prev, a.ldh, a.w_hh, a.ldw_hh, anh, T_{}, F_{}, nullptr, 0);
    }
    if (!EARLY) cell_epilogue_request<RT>(a, m0, u0, lane, pre);
    cell_epilogue_items<RT, HAS_TAB, HAS_RB>(a, m0, u0, lane, tr + wave * 4 * 320, tok, arz, anx, anh, pre);
}

template <int RT, int PF, bool HAS_TAB, bool HAS_RB>
int launch_cell_wlds_ovl_inst(const CellArgs& a, hipStream_t st) {
    static std::atomic<bool> attr_set[32];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    auto k = gru_cell_wlds_ovl_kernel<RT, PF, HAS_TAB, HAS_RB>;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int tiles = ((a.B + 64 * RT - 1) / (64 * RT)) * (a.H / 16);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), (size_t)48 * 512 * 4 + 4 * 4 * 320 * 4, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

static bool cell_wlds_ovl_ok(const CellArgs& a) { return a.H == 512 && (!a.x || a.K1 == 512); }

template <int RT, int PF>
int launch_cell_wlds_ovl(const CellArgs& a, hipStream_t st) {
    const bool tab = a.gx_table != nullptr, rb = a.gx_rowbias != nullptr;
    if (tab && rb) return launch_cell_wlds_ovl_inst<RT, PF, true, true>(a, st);
    if (tab) return launch_cell_wlds_ovl_inst<RT, PF, true, false>(a, st);
    if (rb) return launch_cell_wlds_ovl_inst<RT, PF, false, true>(a, st);
    return launch_cell_wlds_ovl_inst<RT, PF, false, false>(a, st);
}

template <int BM, int WM, int WN>
int launch_cell(const CellArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * (Stage<BM, GC_BK, true, NT>::WORDS + Stage<GC_BN, GC_BK, true, NT>::WORDS) * sizeof(float);
    static std::atomic<bool> attr_set[32];         // write-once per device; setting the attribute twice is harmless
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    auto k = gru_cell_kernel<BM, WM, WN>;
    if (!attr_set[dev].load(std::memory_order_acquire)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set[dev].store(true, std::memory_order_release);
    }
    const int tiles = ((a.B + BM - 1) / BM) * (a.H / 32);
    hipLaunchKernelGGL(k, dim3(tiles), dim3(NT), lds, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

extern "C" int fn_weight_images(const FnWeightImage* jobs, int n_jobs, void* stream) {
    if (!jobs) return FN_E_NULL;
    if (n_jobs <= 0 || n_jobs > WI_MAX_JOBS) return FN_E_COUNT;
    WiArgs a;
    for (int j = 0; j < n_jobs; ++j) {
        const FnWeightImage& d = jobs[j];
        if (!d.src || !d.dst) return FN_E_NULL;
        if (d.rows <= 0 || d.cols <= 0 || d.ld < d.cols || d.kind < 0 || d.kind > 5) return FN_E_SHAPE;
        if ((d.kind == 1 || d.kind == 3) && (d.cols % 32)) return FN_E_SHAPE;
        if ((d.kind == 2 || d.kind == 4) && (d.rows % 32)) return FN_E_SHAPE;
        if (d.kind != 0 && (((uintptr_t)d.dst) & 15)) return FN_E_ALIGN;
        a.job[j] = WiJob{d.src, d.dst, d.rows, d.cols, d.ld, d.kind};
    }
    hipLaunchKernelGGL(weight_images_kernel, dim3(128, n_jobs), dim3(256), 0, (hipStream_t)stream, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

extern "C" int fn_frag3_pack(const float* src, int rows, int K, int ld, void* dst, void* stream) {
    if (!src || !dst) return FN_E_NULL;
    if (rows <= 0 || K <= 0 || (K % 32) != 0 || ld < K) return FN_E_SHAPE;
    if (((uintptr_t)dst) & 15) return FN_E_ALIGN;
    const long total = (long)((rows + 15) & ~15) * (K >> 3);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(frag3_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, rows, K, (long)ld, reinterpret_cast<unsigned*>(dst));
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int launch_pack(const float* src, int rows, int K, long ld, float* dst, hipStream_t st) {
    const long total = (long)((rows + 15) & ~15) * (K >> 2);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(frag_pack_kernel, dim3(blocks), dim3(256), 0, st, src, rows, K, ld, dst);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

namespace {

// ---- launch configuration -------------------------------------------------------------------------------
// (TM, TN, D) = M-tiles per workgroup, N-tiles (backward only), chunks in flight per wave.  Picked by measurement on MI355X
// (profiles/).
struct Cfg { int tm, tn, d; };

template <int TM, int D>
void launch_fwd_ns(const FwdArgs& a, int tiles, hipStream_t st) {
    if (a.n == 1) hipLaunchKernelGGL((gru_fwd_step_kernel<1, TM, D>), dim3(tiles), dim3(NT), 0, st, a);
    else if (a.n <= 3) hipLaunchKernelGGL((gru_fwd_step_kernel<3, TM, D>), dim3(tiles), dim3(NT), 0, st, a);
    else if (a.n == 4) hipLaunchKernelGGL((gru_fwd_step_kernel<4, TM, D>), dim3(tiles), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((gru_fwd_step_kernel<8, TM, D>), dim3(tiles), dim3(NT), 0, st, a);
}
template <int TM, int TN, int D>
void launch_bwd_ns(const BwdArgs& a, int tiles, hipStream_t st) {
    if (a.n == 1) hipLaunchKernelGGL((gru_bwd_step_kernel<1, TM, TN, D>), dim3(tiles), dim3(NT), 0, st, a);
    else if (a.n <= 3) hipLaunchKernelGGL((gru_bwd_step_kernel<3, TM, TN, D>), dim3(tiles), dim3(NT), 0, st, a);
    else if (a.n == 4) hipLaunchKernelGGL((gru_bwd_step_kernel<4, TM, TN, D>), dim3(tiles), dim3(NT), 0, st, a);
    else hipLaunchKernelGGL((gru_bwd_step_kernel<8, TM, TN, D>), dim3(tiles), dim3(NT), 0, st, a);
}

bool launch_fwd(const Cfg& c, const FwdArgs& a, int tiles, hipStream_t st) {
#define FN_F(TM_, D_) if (c.tm == TM_ && c.d == D_) { launch_fwd_ns<TM_, D_>(a, tiles, st); return true; }
    FN_F(1, 2) FN_F(1, 3) FN_F(1, 4) FN_F(2, 2) FN_F(2, 3) FN_F(2, 4) FN_F(4, 1) FN_F(4, 2) FN_F(4, 3)
#undef FN_F
    return false;
}
bool launch_bwd(const Cfg& c, const BwdArgs& a, int tiles, hipStream_t st) {
#define FN_B(TM_, TN_, D_) if (c.tm == TM_ && c.tn == TN_ && c.d == D_) { launch_bwd_ns<TM_, TN_, D_>(a, tiles, st); return true; }
    FN_B(1, 1, 3) FN_B(1, 1, 4) FN_B(2, 1, 2) FN_B(2, 1, 3) FN_B(2, 1, 4) FN_B(2, 2, 2) FN_B(2, 2, 3) FN_B(2, 2, 4)
    FN_B(4, 1, 2) FN_B(4, 1, 3) FN_B(4, 2, 2) FN_B(4, 2, 3) FN_B(4, 2, 4) FN_B(1, 2, 3) FN_B(1, 2, 4)
#undef FN_B
    return false;
}

}  // namespace

extern "C" {

int fn_gru_cell_f32(const FnGruCell* c, void* stream) {
    if (!c || !c->h_prev || !c->w_hh || !c->b_hh || !c->h_out) return FN_E_NULL;
    if (c->B <= 0 || c->H <= 0 || (c->H % 32) != 0 || c->ldh < c->H || c->ldo < c->H || c->ldw_hh < c->H) return FN_E_SHAPE;
    if (c->x && (!c->w_ih || c->K1 <= 0 || c->ldx < c->K1 || c->ldw_ih < c->K1)) return FN_E_SHAPE;
    if (c->gx_table && c->idx && c->idx_ld <= 0) return FN_E_SHAPE;
    if (c->h_out == c->h_prev) return FN_E_SHAPE;              // other workgroups still read the old state
    CellArgs a;
    a.x = c->x; a.ldx = c->ldx; a.K1 = c->x ? c->K1 : 0; a.w_ih = c->w_ih; a.ldw_ih = c->ldw_ih;
    a.gx_table = c->gx_table; a.idx = c->idx; a.idx_ld = c->idx_ld; a.tok_const = c->start_token; a.gx_rowbias = c->gx_rowbias;
    a.h_prev = c->h_prev; a.ldh = c->ldh; a.w_hh = c->w_hh; a.ldw_hh = c->ldw_hh; a.b_ih = c->b_ih; a.b_hh = c->b_hh;
    a.h_out = c->h_out; a.ldo = c->ldo; a.B = c->B; a.H = c->H;
    a.best = reinterpret_cast<const unsigned long long*>(c->idx_best); a.best_v = c->best_v;
    if (c->idx_best && (c->best_v <= 0 || !c->gx_table)) return FN_E_SHAPE;
    // variant bit 14 (as in FnGruFwd): the cell on the bf16 MFMA with exact triple splits where the shape allows it (else the fp32 cells below, as if the bit were clear)
    const int variant = c->variant & ~0x4000;
    if ((c->variant & 0x4000) && cell_x6_ok(a)) return launch_cell_x6(a, (hipStream_t)stream);
    // measured (scratch/prof_decode_cells.sh, us per token of the tokens-only decode, profiles/r04_decode_cells_lds_free.txt): the form with the weight slice in
    // LDS wants ONE workgroup per CU: 64 RT rows x 16 units with RT = ceil(rows / 512) - 1024 rows 71.5 (staged 82, LDS-free 77), 1152-1536 rows 88-89 (LDS-free 101-102);
    // at 2048 rows (RT = 4) it is behind the LDS-free 128-row form (108 against 103)
    if (variant == 0 && c->B > 512 && c->B <= 2048 && cell_direct_ok(a) && cell_wlds_ovl_ok(a)) {
        // K1 = H = 512: the slice fills under the K loops - 640-1024 rows 60 (71), 1280-1536 rows 78-80 (88), 2048 rows 99.8 (102.9 LDS-free)
        if (c->B <= 1024) return launch_cell_wlds_ovl<2, 2>(a, (hipStream_t)stream);
        if (c->B <= 1536) return launch_cell_wlds_ovl<3, 2>(a, (hipStream_t)stream);
        return launch_cell_wlds_ovl<4, 2>(a, (hipStream_t)stream);
    }
    if (variant == 0 && c->B > 512 && cell_direct_ok(a)) {
        if (c->B > 1536 || !cell_wlds_ok(a)) return c->B > 1024 ? launch_cell_direct<4, 2>(a, (hipStream_t)stream) : launch_cell_direct<2, 4>(a, (hipStream_t)stream);
        return c->B > 1024 ? launch_cell_wlds<3, 2>(a, (hipStream_t)stream) : launch_cell_wlds<2, 4>(a, (hipStream_t)stream);
    }
    switch (variant) {                                  // tuning / tests: the staged forms agree bit for bit, the LDS-free forms 4-7 among themselves (another k order)
        case 15: if (cell_direct_ok(a) && cell_wlds_ovl_ok(a)) return launch_cell_wlds_ovl<2, 4>(a, (hipStream_t)stream); break;
        case 16: if (cell_direct_ok(a) && cell_wlds_ovl_ok(a)) return launch_cell_wlds_ovl<3, 2>(a, (hipStream_t)stream); break;
        case 17: if (cell_direct_ok(a) && cell_wlds_ovl_ok(a)) return launch_cell_wlds_ovl<4, 2>(a, (hipStream_t)stream); break;
        case 18: if (cell_direct_ok(a) && cell_wlds_ovl_ok(a)) return launch_cell_wlds_ovl<2, 2>(a, (hipStream_t)stream); break;
        case 13: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<3, 4>(a, (hipStream_t)stream); break;
        case 14: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<3, 2>(a, (hipStream_t)stream); break;
        case 9: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<4, 2>(a, (hipStream_t)stream); break;      // (<4, 4> needs AGPR copies: not built)
        case 10: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<2, 4>(a, (hipStream_t)stream); break;
        case 11: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<4, 2>(a, (hipStream_t)stream); break;
        case 12: if (cell_direct_ok(a) && cell_wlds_ok(a)) return launch_cell_wlds<2, 8>(a, (hipStream_t)stream); break;
        case 4: if (cell_direct_ok(a)) return launch_cell_direct<4, 1>(a, (hipStream_t)stream); break;
        case 5: if (cell_direct_ok(a)) return launch_cell_direct<2, 4>(a, (hipStream_t)stream); break;
        case 6: if (cell_direct_ok(a)) return launch_cell_direct<4, 2>(a, (hipStream_t)stream); break;
        case 7: if (cell_direct_ok(a)) return launch_cell_direct<2, 2>(a, (hipStream_t)stream); break;
        default: break;
    }
    switch (variant) {
        case 1: return launch_cell<128, 4, 1>(a, (hipStream_t)stream);
        case 2: return launch_cell<128, 2, 2>(a, (hipStream_t)stream);
        case 3: return launch_cell<64, 4, 1>(a, (hipStream_t)stream);
        default: return launch_cell<64, 2, 2>(a, (hipStream_t)stream);
    }
}

static bool fn_out_argmax_force_direct = getenv("FN_OUT_ARGMAX_DIRECT") != nullptr;     // measurements: the LDS-free form
int fn_out_argmax_f32(const float* h, int ldh, const float* W, int ldw, const float* bias, int B, int V, int K, uint64_t* best, void* stream) {
    if (!h || !W || !bias || !best) return FN_E_NULL;
    if (B <= 0 || V <= 0 || K <= 0 || (K % 16) != 0 || ldh < K || ldw < K || (ldh & 3) || (ldw & 3)) return FN_E_SHAPE;
    if ((long)B * ldh * 4 >= (1L << 32) || (long)V * ldw * 4 >= (1L << 32)) return FN_E_SHAPE;
    if (((((uintptr_t)h) | ((uintptr_t)W)) & 15) || (((uintptr_t)best) & 7)) return FN_E_ALIGN;
    const size_t lds = (size_t)48 * K * sizeof(float);
    if (lds <= (size_t)128 * 1024 && !fn_out_argmax_force_direct) {          // weight slice of a workgroup resident in LDS
        static std::atomic<bool> attr_set[2][32];
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
        const bool eight = (K % 32) == 0 && !getenv("FN_OUT_ARGMAX_4WAVES");
        auto k = eight ? out_argmax_lds8_kernel<4> : out_argmax_lds_kernel<8>;
        if (!attr_set[eight][dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set[eight][dev].store(true, std::memory_order_release);
        }
        const int grid = ((B + 63) / 64) * ((V + 47) / 48);
        hipLaunchKernelGGL(k, dim3(grid), dim3(eight ? 2 * NT : NT), lds + (eight ? 4 * 3 * 64 * 16 : 0), (hipStream_t)stream, h, (long)ldh, W, (long)ldw, bias, B, V, K,
                           reinterpret_cast<unsigned long long*>(best));
        FN_CHECK_LAUNCH();
        return FN_OK;
    }
    const int grid = ((B + 31) / 32) * ((V + 95) / 96);
    hipLaunchKernelGGL((out_argmax_direct_kernel<4>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, h, (long)ldh, W, (long)ldw, bias, B, V, K,
                       reinterpret_cast<unsigned long long*>(best));
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_best_tokens(const uint64_t* best, int steps, int B, int V, int32_t* tokens, int tok_ld, void* stream) {
    if (!best || !tokens) return FN_E_NULL;
    if (steps <= 0 || B <= 0 || V <= 0 || tok_ld < steps) return FN_E_SHAPE;
    const long n = (long)steps * B;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(best_tokens_kernel, dim3((unsigned)(blocks < 2048 ? blocks : 2048)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const unsigned long long*>(best), n, B, V, tokens, (long)tok_ld);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

size_t fn_gru_gates_floats(int B, int H) { return (size_t)4 * H * (((size_t)B + 15) / 16 * 16); }

size_t fn_frag_floats(int rows, int K) { return (size_t)((rows + 15) / 16 * 16) * K; }

int fn_frag_pack(const float* src, int rows, int K, int ld, float* dst, void* stream) {
    if (!src || !dst) return FN_E_NULL;
    if (rows <= 0 || K <= 0 || (K % 32) != 0 || ld < K) return FN_E_SHAPE;
    if (((uintptr_t)dst) & 15) return FN_E_ALIGN;
    return launch_pack(src, rows, K, ld, dst, (hipStream_t)stream);
}

int fn_gru_seq_fwd(const FnGruFwd* scans, int n_scans, void* stream) {
    if (!scans) return FN_E_NULL;
    if (n_scans <= 0 || n_scans > FN_MAX_SCANS) return FN_E_COUNT;
    int Tmax = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruFwd& d = scans[s];
        if (!d.w_hh_frag || !d.b_hh || !d.h_all || !d.frag_ws) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 32) != 0) return FN_E_SHAPE;
        if (d.gx_table && !d.idx) return FN_E_NULL;
        if ((((uintptr_t)d.w_hh_frag) | ((uintptr_t)d.frag_ws) | ((uintptr_t)d.h0_frag) | ((uintptr_t)d.h_last_frag)) & 15) return FN_E_ALIGN;
        if (d.h0_frag && !d.h0) return FN_E_NULL;            // the gate epilogue reads the row-major state
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    hipStream_t st = (hipStream_t)stream;
    {   // weight-stationary single launch when the configuration fits one workgroup per CU (gru_persist.hip)
        const int rc = fn_gru_fwd_persist(scans, n_scans, st);
        if (rc != FN_PERSIST_NA) return rc;
        if (scans[0].variant & 0x4000) return FN_E_UNSUPPORTED;      // the per-step kernels below do not read triple images
    }
    for (int s = 0; s < n_scans; ++s)          // initial states -> fragment-major (slot 0 of the ping-pong scratch)
        if (scans[s].h0 && !scans[s].h0_frag) {
            const int rc = launch_pack(scans[s].h0, scans[s].B, scans[s].H, scans[s].H, scans[s].frag_ws, st);
            if (rc != FN_OK) return rc;
        }
    for (int p = 0; p < Tmax; ++p) {
        // row-tile height: 64 rows when the launch already fills the chip, 32 otherwise (more workgroups)
        long big_tiles = 0;
        for (int s = 0; s < n_scans; ++s)
            if (p < scans[s].T) big_tiles += (long)((scans[s].B + 63) / 64) * (scans[s].H / 16);
        Cfg cfg = big_tiles >= 256 ? Cfg{4, 0, 1} : Cfg{2, 0, 3};
        const int bm = 16 * cfg.tm;
        FwdArgs a;
        a.n = 0;
        int tiles = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruFwd& d = scans[s];
            if (p >= d.T) continue;
            FwdStep& f = a.s[a.n++];
            const long BH = (long)d.B * d.H;
            const long FS = (long)fn_frag_floats(d.B, d.H);
            f.w_frag = d.w_hh_frag; f.b_hh = d.b_hh; f.b_ih = d.b_ih;
            f.h_prev = p == 0 ? d.h0 : d.h_all + (p - 1) * BH;
            f.hf_in = (p == 0 && d.h0_frag) ? d.h0_frag : d.frag_ws + (p & 1) * FS;
            f.hf_out = p + 1 < d.T ? d.frag_ws + ((p + 1) & 1) * FS : d.h_last_frag;
            f.h_out = d.h_all + p * BH;
            f.gates = d.gates ? d.gates + (long)p * fn_gru_gates_floats(d.B, d.H) : nullptr;
            f.gx_dense = d.gx_dense ? d.gx_dense + p * 3 * BH : nullptr;
            f.gx_table = d.gx_table;
            f.gx_rowbias = d.gx_rowbias;
            const int tau = (d.reverse ? d.T - 1 - p : p) + d.idx_shift;
            f.idx = (d.gx_table && tau >= 0) ? d.idx + tau : nullptr;
            f.idx_ld = d.idx_ld;
            f.tok_const = d.start_token;
            f.B = d.B; f.H = d.H;
            f.ntm = (d.B + bm - 1) / bm;
            f.tile0 = tiles;
            tiles += f.ntm * (d.H / 16);
        }
        a.total = tiles;
        if (!launch_fwd(cfg, a, tiles, st)) return FN_E_SHAPE;
        FN_CHECK_LAUNCH();
    }
    return FN_OK;
}

int fn_gru_seq_bwd(const FnGruBwd* scans, int n_scans, void* stream) {
    if (!scans) return FN_E_NULL;
    if (n_scans <= 0 || n_scans > FN_MAX_SCANS) return FN_E_COUNT;
    int Tmax = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruBwd& d = scans[s];
        if (!d.w_hh_t_frag || !d.h_all || !d.gates || !d.dgx_all || !d.dghn_all || !d.scratch || !d.frag_ws) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 32) != 0) return FN_E_SHAPE;
        if ((((uintptr_t)d.w_hh_t_frag) | ((uintptr_t)d.frag_ws)) & 15) return FN_E_ALIGN;
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    hipStream_t st = (hipStream_t)stream;
    {
        const int rc = fn_gru_bwd_persist(scans, n_scans, st);
        if (rc != FN_PERSIST_NA) return rc;
        if (scans[0].variant & 0x4000) return FN_E_UNSUPPORTED;      // the per-step kernels below do not read triple images
    }
    for (int it = 0; it <= Tmax; ++it) {
        long big_tiles = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruBwd& d = scans[s];
            if (it > d.T || (it == d.T && !d.dh0)) continue;
            big_tiles += (long)((d.B + 63) / 64) * ((d.H + 31) / 32);
        }
        if (big_tiles == 0) continue;
        Cfg cfg = big_tiles >= 192 ? Cfg{4, 1, 2} : Cfg{2, 1, 3};
        const int bm = 16 * cfg.tm, bn = 16 * cfg.tn;
        BwdArgs a;
        a.n = 0;
        int tiles = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruBwd& d = scans[s];
            if (it > d.T || (it == d.T && !d.dh0)) continue;
            BwdStep& f = a.s[a.n++];
            const long BH = (long)d.B * d.H;
            const long GS = (long)fn_gru_gates_floats(d.B, d.H);
            const int q = d.T - 1 - it;                    // step whose gate backward runs now (-1 on the last)
            const long FS3 = (long)fn_frag_floats(d.B, 3 * d.H);
            f.wt_frag = d.w_hh_t_frag;
            if (it > 0) {
                f.df_in = d.frag_ws + ((it - 1) & 1) * FS3;
                f.dh_in = d.scratch;
            } else {
                f.df_in = nullptr;
                f.dh_in = d.dh_last;
            }
            f.df_out = q >= 0 ? d.frag_ws + (it & 1) * FS3 : nullptr;
            f.dh_ext = (d.dh_ext && q >= 0) ? d.dh_ext + (long)q * BH : nullptr;
            if (q >= 0) {
                f.gates_q = d.gates + (long)q * GS;
                f.hprev_q = q == 0 ? d.h0 : d.h_all + (long)(q - 1) * BH;
                f.dgx_q = d.dgx_all + (long)q * 3 * BH;
                f.dghn_q = d.dghn_all + (long)q * BH;
            } else {
                f.gates_q = nullptr; f.hprev_q = nullptr; f.dgx_q = nullptr; f.dghn_q = nullptr;
            }
            f.dhz_out = d.scratch;
            f.rowsum = d.dgx_rowsum;
            f.rowsum_n = d.dghn_rowsum;
            f.dh0_out = d.dh0;
            f.B = d.B; f.H = d.H;
            f.ntm = (d.B + bm - 1) / bm;
            f.tile0 = tiles;
            tiles += f.ntm * ((d.H + bn - 1) / bn);
        }
        a.total = tiles;
        if (!launch_bwd(cfg, a, tiles, st)) return FN_E_SHAPE;
        FN_CHECK_LAUNCH();
    }
    return FN_OK;
}

}  // extern "C"
