// gru_persist.hip - weight-stationary GRU scans: ONE launch runs all T time steps of up to FN_MAX_SCANS scans
// (encoders gmm_model.py:84,89; sub-decoders :109,114; decoder cells :131-136).
//
// Why: the per-step kernels of gru.hip fetch W_hh (3 MB per scan) again on every step - 23 MB of HBM/MALL reads per 4-scan step
// launch in the PMC counters, the per-XCD L2s do not keep it across kernel boundaries - and pay a dependent-launch boundary
// (~2.7 us) per step: 24.5 us per encoder step against 10.2 us of MFMA time.  One launch per scan: 17.3 us (profiles/).
//
// Decomposition (H = 512: 32 unit slices x 8 row groups = 256 workgroups = one per CU):
//   * workgroup (g, j): row group g = (scan, block of RPW batch rows), slice j = hidden units [16j, 16j+16) for all of
//     r, z, n.  Its W_hh slice (48 rows x H, 96 KB at H = 512) is copied into LDS ONCE, in MFMA B-fragment order;
//   * every step the recurrent operand h_{p-1}[rows of g][0..H) is read from a fragment-major exchange slab in global memory
//     (asm dwordx4 loads kept D chunks in flight) and multiplied against the LDS-resident slice;
//   * the gate epilogue is element-wise; the new state slice goes to h_all (row-major, for the backward pass and the heads)
//     and, write-through (sc1), to the other exchange slab;
//   * the 32 workgroups of a row group then meet at a monotonic arrival counter (cdna_hip_programming.md G16, recipe R1:
//     sc1 payload stores -> every wave drains vmcnt -> barrier -> one relaxed agent-scope atomic; consumer: one lane polls,
//     barrier, sc1 loads).  Block ids are dealt so that a row group sits on ONE XCD when there are 8 groups (speed only,
//     correctness never depends on placement).
//   * every spin is bounded: on timeout (or when another workgroup has timed out) the workgroup raises err and leaves.
#include <atomic>
#include <type_traits>

#include "gru_layout.h"
#include "kloop_asm.h"
#include "kloop2_asm.h"
#include "kloop3_asm.h"
#include "kloop4_asm.h"

#ifdef FN_TIMING
__device__ unsigned long long fn_pdbg[8 * 8];
#define FN_PSTAMP(k)                                                                                         \
    do {                                                                                                     \
        if (blockIdx.x < 8 && threadIdx.x == 0 && p == 10) fn_pdbg[blockIdx.x * 8 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
__device__ unsigned long long fn_pcnt[8];           // event counters of workgroup 0, thread 0 (ping-pong scans: [0] = counter polls that fell short)
#define FN_PCOUNT(k)                                                      \
    do {                                                                  \
        if (blockIdx.x == 0 && threadIdx.x == 0) fn_pcnt[(k)] += 1;       \
    } while (0)
extern "C" int fn_pdbg_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fn_pdbg), sizeof(unsigned long long) * 64);
}
extern "C" int fn_pcnt_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fn_pcnt), sizeof(unsigned long long) * 8);
}
#else
#define FN_PSTAMP(k)
#define FN_PCOUNT(k)
#endif

namespace {

constexpr int NT = 256;

// LDS byte address of a pointer into the workgroup's shared memory (what ds_read / ds_write take)
FN_DEVINL unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

struct PScan {
    const float* w_frag;
    const float* b_hh;
    const float* b_ih;
    const float* h0;
    const float* gx_dense;
    const float* gx_table;
    const int* idx;
    const float* gx_rowbias;
    float* h_all;
    float* gates;
    float* xf;            // 2 exchange slabs, fragment-major
    const float* h0f;     // optional: the initial state already in the slab layout (the previous chunk's hlf) - no packing launch
    float* hlf;           // optional: the final state in the slab layout
    int idx_ld, idx_shift, start_token, reverse;
    int B, T;
    int group0;           // first row group of this scan
};
struct PArgs {
    PScan s[FN_MAX_SCANS];
    int n, ngroups, H;
    int no_hand;          // tuning / tests: the compiler-scheduled K loop instead of kloop_asm.h
    int spread;           // tests: deal the block ids so that every row group is spread over all XCDs (variant bit 12; see block_group)
    u32* sync;            // [ngroups * 32] arrival counters (one per 128-byte line), zero at launch
    u32* err;             // sticky error word
};
constexpr int FN_MAX_GROUPS = 64;

// Which (row group, slice) a workgroup is.  Default: group = id mod groups - with 8 (or 16) groups every row group sits on ONE XCD (workgroups are
// dealt to the XCDs round robin by id), its exchange slab then lives in that XCD's L2.  That is speed only: the hand-over protocol is agent-scope
// (sc1 stores and loads, agent-scope counters).  spread (FnGruFwd / FnGruBwd.variant bit 12) deals consecutive ids to the slices of one group, so
// that every group is spread over all XCDs: the test that correctness does not depend on the placement.
FN_DEVINL int block_group(int ngroups, int spread) { return spread ? (int)blockIdx.x / ((int)gridDim.x / ngroups) : (int)blockIdx.x % ngroups; }
FN_DEVINL int block_slice(int ngroups, int spread) { return spread ? (int)blockIdx.x % ((int)gridDim.x / ngroups) : (int)blockIdx.x / ngroups; }


// waves = WM (row blocks of MT tiles) x WK (K split); rows per workgroup RPW = 16 * WM * MT
template <int WM, int WK, int MT, int D>
__global__ __launch_bounds__(NT) void gru_fwd_persist_kernel(const PArgs args) {
    static_assert(WM * WK == 4 && (D % 2) == 0, "4 waves; ring depth even");
    constexpr int EM = WM * MT;                      // row tiles per workgroup = epilogue rows per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = args.H, nk = H >> 5;
    float* wl = smem;                                // [3][nk][2][64][4]  W_hh slice, B-fragment order
    float* red = smem + 3 * H * 16;                  // [WK][EM][3][RT]    accumulator exchange
    volatile int& dead = *reinterpret_cast<volatile int*>(red + WK * EM * 3 * RT);   // all LDS is dynamic (16-byte aligned base)

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const PScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (16 * EM), hh0 = slice * 16;
    const int nrt = (B + 15) >> 4;
    const int nslices = H >> 4;
    const long FS = (long)nrt * 16 * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wk = wave / WM;
    u32* counter = args.sync + g * 32;
    u32* err = args.err;

    // ---- one-time: W_hh slice -> LDS (already in fragment order in global memory) -------------------------------------
    if (tid == 0) dead = 0;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const float4* src = reinterpret_cast<const float4*>(S.w_frag + (long)(q * nslices + slice) * nk * 512);
        float4* dst = reinterpret_cast<float4*>(wl + (long)q * nk * 512);
        for (int i = tid; i < nk * 128; i += NT) dst[i] = src[i];
    }

    // ---- one-time: per-thread epilogue constants.  Epilogue item = (row, 4 consecutive units): every global access of
    //      the epilogue is a 16-byte vector (scalar sc1 stores cost ~6x per byte, MI355X_MICROARCH.md price list) -------
    constexpr int NI = (EM * 64 + NT - 1) / NT;      // items per thread
    int ib[NI], iu4[NI], icoff[NI];
    bool iact[NI];
    f32x4 bh[3], bi[3], e_rb[NI][3], hp[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int item = tid + NT * i;
        const int rl = item >> 2;                    // row inside the workgroup's row block
        iu4[i] = item & 3;
        iact[i] = item < EM * 64 && m0 + rl < B;
        ib[i] = min(m0 + rl, B - 1);
        // accumulator tile (rl >> 4) in padded MFMA C layout: lane' = ((r&15)>>2)*16 + unit, reg = r & 3, +4 floats per 64
        icoff[i] = (min(rl >> 4, EM - 1) * 3) * RT + ((rl & 15) >> 2) * 68 + iu4[i] * 16 + (rl & 3);
    }
    const int jj0 = hh0 + 4 * (tid & 3);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bh[q] = ldv4(S.b_hh + q * H + jj0);
        bi[q] = S.b_ih ? ldv4(S.b_ih + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        hp[i] = S.h0 ? ldv4(S.h0 + (long)ib[i] * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            e_rb[i][q] = S.gx_rowbias ? ldv4(S.gx_rowbias + (long)ib[i] * 3 * H + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    // this wave's K chunks [c0, c0 + nkw) and operand row tiles
    const int c0 = nk * wk / WK, nkw = nk * (wk + 1) / WK - c0;
    long aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = (long)min((m0 >> 4) + wm * MT + m, nrt - 1) * nk * 512 + lane * 4;
    constexpr int NLA = 2 * MT;                      // asm loads per chunk
    // H = 512 with two row tiles per wave (the 128- and 64-row tilings of the training shapes): the hand-placed K loop of kloop_asm.h
    const bool hand = MT == 2 && H == 512 && (nkw == 16 || nkw == 8) && !args.no_hand && ((S.gx_table != nullptr) != (S.gx_dense != nullptr));
    const unsigned h_lp = lds_addr(wl) + c0 * 2048 + lane * 16;
    const unsigned h_red = lds_addr(red) + (((wk * EM + wm * MT) * 3) * RT + lane * 4 + (lane >> 4) * 4) * 4;

    // token of the step about to run (read one step ahead: the table-row address must not wait for it)
    int tokn[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int tau0 = (S.reverse ? T - 1 : 0) + S.idx_shift;
        tokn[i] = (S.gx_table && tau0 >= 0) ? S.idx[(long)ib[i] * S.idx_ld + tau0] : S.start_token;
    }

#pragma unroll 1
    for (int p = 0; p < T; ++p) {
        const bool has_k = p > 0 || S.h0 != nullptr;
        const bool hand_now = MT == 2 && hand && has_k;
        // (a) h-independent epilogue operands of this step: requested inside the hand-placed K loop (only their addresses here), else
        //     up front (their latency then hides under the wait and the K loop)
        f32x4 e_x[NI][3];
        const float* xa[2] = {nullptr, nullptr};
        if (MT == 2 && hand_now) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ii = i < NI ? i : 0;
                xa[i] = S.gx_table ? S.gx_table + (long)tokn[ii] * 3 * H + jj0 + H : S.gx_dense + ((long)p * B + ib[ii]) * 3 * H + jj0 + H;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
#pragma unroll
                for (int q = 0; q < 3; ++q) e_x[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (S.gx_table) {
                    const float* row = S.gx_table + (long)tokn[i] * 3 * H + jj0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) e_x[i][q] = ldv4(row + q * H);
                }
                if (S.gx_dense) {
                    const float* row = S.gx_dense + ((long)p * B + ib[i]) * 3 * H + jj0;
#pragma unroll
                    for (int q = 0; q < 3; ++q) e_x[i][q] += ldv4(row + q * H);
                }
            }
        }

        FN_PSTAMP(0);
        // (b) wait until every slice of this row group has published h_{p-1}
        if (p > 0) {
            if (tid == 0) {
                const u32 target = (u32)nslices * (u32)p;
                u32 spins = 0;
                while (ld_cnt(counter) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0 && (spins > SPIN_LIMIT || ld_cnt(err) != 0)) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = 1;
                        break;
                    }
                }
            }
            __syncthreads();
            if (dead) return;
        }

        FN_PSTAMP(1);
        // (c) gh = h_{p-1} W_hh^T slice
        f32x4 acc[MT][3];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MT == 2 && hand_now) {
            const float* xin = (p == 0 && S.h0f) ? S.h0f : S.xf + (long)(p & 1) * FS;
            // MFMAs, operand ring, weight-fragment reads, the epilogue operands (a) and the accumulator hand-over (d) of this step
            f32x4 ex[2][3];
            if (WK == 1) fn_kloop_fwd_h512_k512(xin + (long)c0 * 512, (unsigned)aoff[0] * 4u, (unsigned)aoff[MT - 1] * 4u, h_lp, h_lp + 65536u, h_red, xa[0], xa[1], ex);
            else fn_kloop_fwd_h512_k256(xin + (long)c0 * 512, (unsigned)aoff[0] * 4u, (unsigned)aoff[MT - 1] * 4u, h_lp, h_lp + 65536u, h_red, xa[0], xa[1], ex);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) e_x[i][q] = ex[i < 2 ? i : 0][q];
        } else if (has_k && nkw > 0) {
            const float* xin = (p == 0 && S.h0f) ? S.h0f : S.xf + (long)(p & 1) * FS;
            f32x4 fa[D][MT][2], fb[2][3][2];
            auto loadA = [&](int set, int it) {
                const long k0 = (long)(c0 + min(it, nkw - 1)) * 512;
#pragma unroll
                for (int m = 0; m < MT; ++m) { gld4_sc1(fa[set][m][0], xin + aoff[m] + k0); gld4_sc1(fa[set][m][1], xin + aoff[m] + k0 + 256); }
            };
            auto loadB = [&](int set, int it) {
                const int c = c0 + min(it, nkw - 1);
#pragma unroll
                for (int n = 0; n < 3; ++n) {
                    fb[set][n][0] = *reinterpret_cast<const f32x4*>(wl + ((long)(n * nk + c) * 2 + 0) * 256 + lane * 4);
                    fb[set][n][1] = *reinterpret_cast<const f32x4*>(wl + ((long)(n * nk + c) * 2 + 1) * 256 + lane * 4);
                }
            };
            auto mma = [&](int set, int bs) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < 3; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(fa[set][m][j >> 2], j & 3), f4at(fb[bs][n][j >> 2], j & 3),
                                                                             acc[m][n], 0, 0, 0);
            };
#pragma unroll
            for (int s = 0; s < D; ++s) loadA(s, s);
            loadB(0, 0);
            const int nmain = nkw / D * D;
            for (int base = 0; base < nmain; base += D) {
#pragma unroll
                for (int uu = 0; uu < D; ++uu) {
                    loadB((uu + 1) & 1, base + uu + 1);
                    fn_wait_vm<NLA * (D - 1)>();
                    mma(uu, uu & 1);
                    loadA(uu, base + uu + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fn_wait_vm<0>();
#pragma unroll
            for (int uu = 0; uu < D; ++uu)
                if (nmain + uu < nkw) {
                    loadB((uu + 1) & 1, nmain + uu + 1);
                    mma(uu, uu & 1);
                }
#pragma unroll
            for (int s = 0; s < D; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) { fn_keep(fa[s][m][0]); fn_keep(fa[s][m][1]); }
        }

        // next step's token
        if (S.gx_table && p + 1 < T) {
            const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
#pragma unroll
            for (int i = 0; i < NI; ++i) tokn[i] = tau1 >= 0 ? S.idx[(long)ib[i] * S.idx_ld + tau1] : S.start_token;
        }

        FN_PSTAMP(2);
        // (d) accumulators -> LDS in MFMA C layout (row = 4*(lane>>4) + reg, col = lane & 15), 4 floats of padding per
        //     16 lanes so that the epilogue's reads (4 row quads x 4 unit quads per wave) hit 64 different banks
        if (!hand_now) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int n = 0; n < 3; ++n)
                    *reinterpret_cast<f32x4*>(red + ((long)((wk * EM + wm * MT + m) * 3 + n)) * RT + lane * 4 + (lane >> 4) * 4) = acc[m][n];
        }
        __syncthreads();

        FN_PSTAMP(3);
        // (e) gates and the new state: thread -> items (row, units 4u4 .. 4u4+3)
        float* h_out = S.h_all + (long)p * B * H;
        float* gt = S.gates ? S.gates + (long)p * 4 * H * nrt * 16 : nullptr;
        float* xout = (p + 1 < T) ? S.xf + (long)((p + 1) & 1) * FS : S.hlf;     // last step: hand-over slab of the next launch (or none)
        f32x4 o_r[NI], o_z[NI], o_n[NI], o_g[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 gh[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    float a = 0.f;
#pragma unroll
                    for (int w = 0; w < WK; ++w) a += red[(long)(w * EM * 3 + q) * RT + icoff[i] + c * 4];
                    gh[q][c] = a + bh[q][c];
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float r, z, n, hn;
                fn_gru_gate((bi[0][c] + e_x[i][0][c]) + e_rb[i][0][c], (bi[1][c] + e_x[i][1][c]) + e_rb[i][1][c], (bi[2][c] + e_x[i][2][c]) + e_rb[i][2][c],
                            gh[0][c], gh[1][c], gh[2][c], hp[i][c], r, z, n, hn);
                hp[i][c] = hn;
                o_r[i][c] = r; o_z[i][c] = z; o_n[i][c] = n; o_g[i][c] = gh[2][c];
            }
            // the exchange slab first: it is all the other workgroups wait for
            if (xout && iact[i]) stv4_sc1(xout + frag_off(ib[i], jj0, nk), hp[i]);
        }

        FN_PSTAMP(4);
        // (f) publish: every wave drains its stores, then ONE lane arrives at the group counter
        if (p + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            FN_PSTAMP(5);
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FN_PSTAMP(6);
        // (g) outputs nobody in this launch waits for
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (!iact[i]) continue;
            stv4(h_out + (long)ib[i] * H + jj0, hp[i]);
            if (gt) {
                stv4(gt + gate_off(ib[i], 0, jj0, nrt), o_r[i]);
                stv4(gt + gate_off(ib[i], 1, jj0, nrt), o_z[i]);
                stv4(gt + gate_off(ib[i], 2, jj0, nrt), o_n[i]);
                stv4(gt + gate_off(ib[i], 3, jj0, nrt), o_g[i]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward scan with EXACT split products on the bf16 MFMA ("bf16 x 6", gemm.hip / DESIGN.md; OPT-IN: FnGruFwd.variant bit 14).  Weights and
// recurrent operand are bf16 TRIPLES (x = hi + mid + lo exactly; fn_frag3_pack image = [row tile 16][k block 32][piece 3][64 lanes][8 bf16]:
// lane (i, g) of a v_mfma_f32_16x16x32_bf16 operand = row i, k values 8 g .. 8 g + 7 of the block): the W_hh slice is split ONCE per optimiser
// step (144 KB of LDS per workgroup at H = 512), the new state is split by the gate epilogue that publishes it (6 instead of 4 bytes per
// value on the exchange slab), so the K loop runs at the pre-split rate - 6 bf16 MFMAs of 16 cycles per (row tile, gate, 32 k) instead
// of 8 fp32 ones of 32 (scratch/scan_x6_kloop.hip: 6.1 against 12.0 us per encoder-shaped step for the loop alone).  Six of the nine exact
// partial products, smallest first; gate arithmetic = fn_gru_gate as everywhere.  Structure of gru_fwd_persist_kernel (one row group per
// workgroup, counter hand-over, no ping-pong): 4 waves x MT row tiles, every wave all of K (no K split, no accumulator reduction); the
// accumulators reach the (row, 4 units) epilogue items through LDS one tile per wave at a time (MT rounds: 144 KB + 13 KB of LDS).
// H = 512, full row groups of 64 MT rows, saved gates; initial state / chunk hand-over as in gru_fwd_persist_kernel (h0f / hlf are triple images).
// ---------------------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
FN_DEVINL float fn_top16(float x) { return __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
FN_DEVINL unsigned fn_pack_top16(float a, float b) { return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u); }
// four fp32 values -> their exact bf16 triples (rounded pieces, fn_rn16), each piece as four packed bf16: what one epilogue item puts on the exchange slab
FN_DEVINL void fn_split3x4(const f32x4& x, u32x2& th, u32x2& tm, u32x2& tl) {
    float hi[4], mi[4], lo[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        hi[c] = fn_rn16(x[c]);
        const float r1 = x[c] - hi[c];
        mi[c] = fn_rn16(r1);
        lo[c] = r1 - mi[c];
    }
    th = (u32x2){fn_pack_top16(hi[0], hi[1]), fn_pack_top16(hi[2], hi[3])};
    tm = (u32x2){fn_pack_top16(mi[0], mi[1]), fn_pack_top16(mi[2], mi[3])};
    tl = (u32x2){fn_pack_top16(lo[0], lo[1]), fn_pack_top16(lo[2], lo[3])};
}
FN_DEVINL void gld4u_sc1(u32x4& dst, const u32x4* p) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory"); }
FN_DEVINL void stv2_sc1(void* p, const u32x2& v) { asm volatile("global_store_dwordx2 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

template <int MT>
__global__ __launch_bounds__(NT) void gru_fwd_x6_kernel(const PArgs args) {
    constexpr int H = 512, NB = 16, nslices = 32, EM = 4 * MT, D = 4;        // D k-blocks of operands in flight
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);                              // [3][NB][3][64] weight triples, B-operand order
    float* red = smem + 3 * NB * 3 * 64 * 4;                                 // [4 waves][3][RT] one accumulator tile per wave
    volatile int& dead = *reinterpret_cast<volatile int*>(red + 4 * 3 * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const PScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (16 * EM), hh0 = slice * 16;
    const int nrt = B >> 4;
    const long FS = (long)nrt * NB * 3 * 64;                                 // u32x4 elements of one exchange slab
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* counter = args.sync + g * 32;
    u32* err = args.err;
    u32x4* xs = reinterpret_cast<u32x4*>(S.xf);

    if (tid == 0) dead = 0;
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const u32x4* src = reinterpret_cast<const u32x4*>(S.w_frag) + (long)(q * nslices + slice) * NB * 3 * 64;
        u32x4* dst = wl + q * NB * 3 * 64;
        for (int i = tid; i < NB * 3 * 64; i += NT) dst[i] = src[i];
    }
    // epilogue item of this thread in round m: row rl16 of this wave's m-th tile, units jj0 .. jj0 + 3
    const int rl16 = lane >> 2, u4 = lane & 3;
    const int jj0 = hh0 + 4 * u4;
    const int icoff = (wave * 3) * RT + (rl16 >> 2) * 68 + u4 * 16 + (rl16 & 3);
    int ib[MT];
    f32x4 bh[3], bi[3], e_rb[MT][3], hp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        ib[m] = m0 + (wave * MT + m) * 16 + rl16;
        hp[m] = S.h0 ? ldv4(S.h0 + (long)ib[m] * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            e_rb[m][q] = S.gx_rowbias ? ldv4(S.gx_rowbias + (long)ib[m] * 3 * H + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bh[q] = ldv4(S.b_hh + q * H + jj0);
        bi[q] = S.b_ih ? ldv4(S.b_ih + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int tokn[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int tau0 = (S.reverse ? T - 1 : 0) + S.idx_shift;
        tokn[m] = (S.gx_table && tau0 >= 0) ? S.idx[(long)ib[m] * S.idx_ld + tau0] : S.start_token;
    }
    // where this thread's four new state values go on the exchange slab: row tile, k block = slice / 2, k group 2 (slice & 1) + u4 / 2, half u4 & 1
    long soff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
        soff[m] = ((((long)(ib[m] >> 4) * NB + (slice >> 1)) * 3) * 64 + (ib[m] & 15) + 16 * (2 * (slice & 1) + (u4 >> 1))) * 16 + (u4 & 1) * 8;
    long aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = ((long)((m0 >> 4) + wave * MT + m) * NB * 3) * 64 + lane;
    __syncthreads();

#pragma unroll 1
    for (int p = 0; p < T; ++p) {
        // (a) input-side pre-activations of this step
        f32x4 e_x[MT][3];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int q = 0; q < 3; ++q) e_x[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (S.gx_table) {
                const float* row = S.gx_table + (long)tokn[m] * 3 * H + jj0;
#pragma unroll
                for (int q = 0; q < 3; ++q) e_x[m][q] = ldv4(row + q * H);
            }
            if (S.gx_dense) {
                const float* row = S.gx_dense + ((long)p * B + ib[m]) * 3 * H + jj0;
#pragma unroll
                for (int q = 0; q < 3; ++q) e_x[m][q] += ldv4(row + q * H);
            }
        }
        // (b) every slice of this row group has published h_{p-1}
        if (p > 0) {
            if (tid == 0) {
                const u32 target = (u32)nslices * (u32)p;
                u32 spins = 0;
                while (ld_cnt(counter) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0 && (spins > SPIN_LIMIT || ld_cnt(err) != 0)) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = 1;
                        break;
                    }
                }
            }
            __syncthreads();
            if (dead) return;
        }
        // (c) gh = h_{p-1} W_hh^T slice: six bf16 MFMAs per (row tile, gate, k block)
        f32x4 acc[MT][3];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (p > 0 || S.h0) {
            // step 0 of a launch with an initial state: its triples come from the previous chunk's hand-over image or were packed into slab 0
            const u32x4* xin = (p == 0 && S.h0f) ? reinterpret_cast<const u32x4*>(S.h0f) : xs + (long)(p & 1) * FS;
            u32x4 a[D][MT][3];
            auto load = [&](int set, int blk) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) gld4u_sc1(a[set][m][pc], xin + aoff[m] + (blk * 3 + pc) * 64);
            };
            auto mma = [&](int set, int blk) {
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const bf16x8 bhh = __builtin_bit_cast(bf16x8, wl[((q * NB + blk) * 3 + 0) * 64 + lane]);
                    const bf16x8 bmm = __builtin_bit_cast(bf16x8, wl[((q * NB + blk) * 3 + 1) * 64 + lane]);
                    const bf16x8 bll = __builtin_bit_cast(bf16x8, wl[((q * NB + blk) * 3 + 2) * 64 + lane]);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, a[set][m][0]), am = __builtin_bit_cast(bf16x8, a[set][m][1]),
                                     al = __builtin_bit_cast(bf16x8, a[set][m][2]);
                        f32x4& c = acc[m][q];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bhh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bll, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bmm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bhh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bmm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bhh, c, 0, 0, 0);
                    }
                }
            };
#pragma unroll
            for (int b = 0; b < D - 1; ++b) load(b, b);
#pragma unroll
            for (int blk = 0; blk < NB; ++blk) {
                if (blk + D - 1 < NB) load((blk + D - 1) % D, blk + D - 1);
                const int behind = (blk + D - 1 < NB ? D - 1 : NB - 1 - blk);       // blocks requested behind blk: 3 MT loads each
                if (behind >= 3) fn_wait_vm<9 * MT>();
                else if (behind == 2) fn_wait_vm<6 * MT>();
                else if (behind == 1) fn_wait_vm<3 * MT>();
                else fn_wait_vm<0>();
                mma(blk % D, blk);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int b = 0; b < D; ++b)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) asm volatile("" ::"v"(a[b][m][pc]));
        }
        if (S.gx_table && p + 1 < T) {
            const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
#pragma unroll
            for (int m = 0; m < MT; ++m) tokn[m] = tau1 >= 0 ? S.idx[(long)ib[m] * S.idx_ld + tau1] : S.start_token;
        }
        // (d) + (e): MT rounds - this wave's m-th accumulator tile to LDS, then the gates of its rows as (row, 4 units) items
        float* h_out = S.h_all + (long)p * B * H;
        float* gt = S.gates + (long)p * 4 * H * nrt * 16;
        char* xout = (p + 1 < T) ? reinterpret_cast<char*>(xs + (long)((p + 1) & 1) * FS) : reinterpret_cast<char*>(S.hlf);    // last step: the next chunk's hand-over image (or none)
        f32x4 o_r[MT], o_z[MT], o_n[MT], o_g[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            // the tile region is private to this wave (its 64 lanes write the tile and read it back as 64 items): no workgroup barrier - LDS
            // operations of one wave complete in order; the asm statements keep the compiler from moving accesses across them
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int q = 0; q < 3; ++q) *reinterpret_cast<f32x4*>(red + (long)(wave * 3 + q) * RT + lane * 4 + (lane >> 4) * 4) = acc[m][q];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            f32x4 gh[3];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) gh[q][c] = red[(long)q * RT + icoff + c * 4] + bh[q][c];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float r, z, n, hn;
                fn_gru_gate((bi[0][c] + e_x[m][0][c]) + e_rb[m][0][c], (bi[1][c] + e_x[m][1][c]) + e_rb[m][1][c], (bi[2][c] + e_x[m][2][c]) + e_rb[m][2][c],
                            gh[0][c], gh[1][c], gh[2][c], hp[m][c], r, z, n, hn);
                hp[m][c] = hn;
                o_r[m][c] = r; o_z[m][c] = z; o_n[m][c] = n; o_g[m][c] = gh[2][c];
            }
            if (xout) {                                  // the new state as bf16 triples, straight into the next step's operand layout
                u32x2 th, tm, tl;
                fn_split3x4(hp[m], th, tm, tl);
                stv2_sc1(xout + soff[m], th);
                stv2_sc1(xout + soff[m] + 1024, tm);
                stv2_sc1(xout + soff[m] + 2048, tl);
            }
        }
        // (f) publish: every wave drains its stores, then ONE lane arrives at the group counter
        if (p + 1 < T) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (g) outputs nobody in this launch waits for
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            stv4(h_out + (long)ib[m] * H + jj0, hp[m]);
            stv4(gt + gate_off(ib[m], 0, jj0, nrt), o_r[m]);
            stv4(gt + gate_off(ib[m], 1, jj0, nrt), o_z[m]);
            stv4(gt + gate_off(ib[m], 2, jj0, nrt), o_n[m]);
            stv4(gt + gate_off(ib[m], 3, jj0, nrt), o_g[m]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// backward: dh_q = [dr' | dz' | dn' r]_{q+1} W_hh + dh_q z_{q+1} (carried in registers) + dh_ext[q]; gate backward of step q.
// Workgroup (g, j) owns dh columns [16j, 16j+16): its W_hh^T slice (16 rows x 3H, 96 KB at H = 512) stays in LDS; the
// operand exchanged between the slices is the [rows][3H] pre-activation gradient (3x the forward's volume).
// ---------------------------------------------------------------------------------------------------------------
struct QScan {
    const float* wt_frag;
    const float* h0;
    const float* h_all;
    const float* gates;
    const float* dh_last;
    const float* dh_ext;
    float* dgx_all;
    float* dghn_all;
    float* dh0;
    float* rowsum;
    float* rowsum_n;
    float* xf;            // 2 exchange slabs [rows][3H], fragment-major
    int B, T;
    int group0;
};
struct QArgs {
    QScan s[FN_MAX_SCANS];
    int n, ngroups, H;
    int no_hand;
    int spread;           // as in PArgs
    u32* sync;
    u32* err;
};

template <int WM, int WK, int MT, int D>
__global__ __launch_bounds__(NT) void gru_bwd_persist_kernel(const QArgs args) {
    static_assert(WM * WK == 4 && (D % 2) == 0, "4 waves; ring depth even");
    constexpr int EM = WM * MT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = args.H, nk3 = (3 * H) >> 5;
    float* wl = smem;                                // [nk3][2][64][4]  W_hh^T slice, B-fragment order
    float* red = smem + 3 * H * 16;                  // [2][WK][EM][RT]  (second plane: the hand-placed K loop's odd-k accumulators)
    volatile int& dead = *reinterpret_cast<volatile int*>(red + 2 * WK * EM * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const QScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (16 * EM), hh0 = slice * 16;
    const int nrt = (B + 15) >> 4;
    const int nslices = H >> 4;
    const long FS3 = (long)nrt * 16 * 3 * H;
    const long BH = (long)B * H;
    const long GS = (long)4 * H * nrt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wk = wave / WM;
    u32* counter = args.sync + g * 32;
    u32* err = args.err;

    if (tid == 0) dead = 0;
    {
        const float4* src = reinterpret_cast<const float4*>(S.wt_frag + (long)slice * nk3 * 512);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < nk3 * 128; i += NT) dst[i] = src[i];
    }

    constexpr int NI = (EM * 64 + NT - 1) / NT;
    int ib[NI], icoff[NI];
    bool iact[NI];
    f32x4 carry[NI], rs[NI][3], rsn[NI];
    const int jj0 = hh0 + 4 * (tid & 3);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int item = tid + NT * i;
        const int rl = item >> 2;
        iact[i] = item < EM * 64 && m0 + rl < B;
        ib[i] = min(m0 + rl, B - 1);
        icoff[i] = min(rl >> 4, EM - 1) * RT + ((rl & 15) >> 2) * 68 + (item & 3) * 16 + (rl & 3);
        carry[i] = S.dh_last ? ldv4(S.dh_last + (long)ib[i] * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        rsn[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q) rs[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    const int c0 = nk3 * wk / WK, nkw = nk3 * (wk + 1) / WK - c0;
    long aoff[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) aoff[m] = (long)min((m0 >> 4) + wm * MT + m, nrt - 1) * nk3 * 512 + lane * 4;
    constexpr int NLA = 2 * MT;
    const int iters = T + (S.dh0 ? 1 : 0);
    const bool hand = MT == 2 && H == 512 && (nkw == 48 || nkw == 24) && !args.no_hand;      // kloop_asm.h
    const unsigned h_lp = lds_addr(wl) + c0 * 2048 + lane * 16;
    const unsigned h_red = lds_addr(red) + ((wk * EM + wm * MT) * RT + lane * 4 + (lane >> 4) * 4) * 4;
    static_assert(WK * EM == 8 || MT != 2, "kloop_asm.h: the second accumulator plane sits 8 tiles behind the first");

#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int q = T - 1 - it;                     // step whose gate backward runs now (-1: only dh0 is left)
        const int p = it;                             // FN_PSTAMP
        (void)p;
        FN_PSTAMP(0);
        // (a) operands that do not depend on the exchange: requested inside the hand-placed K loop (only their addresses here), else up front
        const bool hand_now = MT == 2 && hand && it > 0;
        f32x4 g_r[NI], g_z[NI], g_n[NI], g_hn[NI], hpv[NI], ext[NI];
        const float *ga[2], *ha[2], *xa[2];
        const int qc = q > 0 ? q : 0;
        const bool hzero = q < 0 || (q == 0 && !S.h0), xzero = q < 0 || !S.dh_ext;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ii = i < NI ? i : 0;
            const long ro = (long)ib[ii] * H + jj0;
            ga[i] = S.gates + (long)qc * GS + gate_off(ib[ii], 0, jj0, nrt);
            ha[i] = qc > 0 ? S.h_all + (long)(qc - 1) * BH + ro : (S.h0 ? S.h0 + ro : S.h_all + ro);      // unused values still come from a legal address
            xa[i] = S.dh_ext ? S.dh_ext + (long)qc * BH + ro : S.h_all + ro;
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            g_r[i] = g_z[i] = g_n[i] = g_hn[i] = hpv[i] = ext[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (q >= 0 && !hand_now) {
                const float* gq = S.gates + (long)q * GS;
                g_r[i] = ldv4(gq + gate_off(ib[i], 0, jj0, nrt));
                g_z[i] = ldv4(gq + gate_off(ib[i], 1, jj0, nrt));
                g_n[i] = ldv4(gq + gate_off(ib[i], 2, jj0, nrt));
                g_hn[i] = ldv4(gq + gate_off(ib[i], 3, jj0, nrt));
                if (q > 0) hpv[i] = ldv4(S.h_all + (long)(q - 1) * BH + (long)ib[i] * H + jj0);
                else if (S.h0) hpv[i] = ldv4(S.h0 + (long)ib[i] * H + jj0);
                if (S.dh_ext) ext[i] = ldv4(S.dh_ext + (long)q * BH + (long)ib[i] * H + jj0);
            }
        }

        FN_PSTAMP(1);
        // (b) wait for every slice's gradient slab of the previous iteration
        if (it > 0) {
            if (tid == 0) {
                const u32 target = (u32)nslices * (u32)it;
                u32 spins = 0;
                while (ld_cnt(counter) < target) {
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 63u) == 0 && (spins > SPIN_LIMIT || ld_cnt(err) != 0)) {
                        __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        dead = 1;
                        break;
                    }
                }
            }
            __syncthreads();
            if (dead) return;
        }

        FN_PSTAMP(2);
        // (c) dh partial = df_{q+1} W_hh (columns of this slice)
        f32x4 acc[MT][2];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m][0] = acc[m][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (MT == 2 && hand_now) {
            const float* xin = S.xf + (long)((it - 1) & 1) * FS3;
            f32x4 gt[2][4], hp2[2], xt2[2];
            if (WK == 1) fn_kloop_bwd_h512_k1536(xin + (long)c0 * 512, (unsigned)aoff[0] * 4u, (unsigned)aoff[MT - 1] * 4u, h_lp, h_red, ga[0], ga[1], ha[0], ha[1], xa[0], xa[1],
                                                 gt, hp2, xt2);
            else fn_kloop_bwd_h512_k768(xin + (long)c0 * 512, (unsigned)aoff[0] * 4u, (unsigned)aoff[MT - 1] * 4u, h_lp, h_red, ga[0], ga[1], ha[0], ha[1], xa[0], xa[1],
                                        gt, hp2, xt2);
            const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int ii = i < 2 ? i : 0;
                g_r[i] = gt[ii][0]; g_z[i] = gt[ii][1]; g_n[i] = gt[ii][2]; g_hn[i] = gt[ii][3];
                hpv[i] = hzero ? zero4 : hp2[ii];
                ext[i] = xzero ? zero4 : xt2[ii];
            }
        } else if (it > 0 && nkw > 0) {
            const float* xin = S.xf + (long)((it - 1) & 1) * FS3;
            f32x4 fa[D][MT][2], fb[2][2];
            auto loadA = [&](int set, int k) {
                const long k0 = (long)(c0 + min(k, nkw - 1)) * 512;
#pragma unroll
                for (int m = 0; m < MT; ++m) { gld4_sc1(fa[set][m][0], xin + aoff[m] + k0); gld4_sc1(fa[set][m][1], xin + aoff[m] + k0 + 256); }
            };
            auto loadB = [&](int set, int k) {
                const int c = c0 + min(k, nkw - 1);
                fb[set][0] = *reinterpret_cast<const f32x4*>(wl + ((long)c * 2 + 0) * 256 + lane * 4);
                fb[set][1] = *reinterpret_cast<const f32x4*>(wl + ((long)c * 2 + 1) * 256 + lane * 4);
            };
            auto mma = [&](int set, int bs) {
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][j & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(fa[set][m][j >> 2], j & 3), f4at(fb[bs][j >> 2], j & 3),
                                                                             acc[m][j & 1], 0, 0, 0);
            };
#pragma unroll
            for (int s = 0; s < D; ++s) loadA(s, s);
            loadB(0, 0);
            const int nmain = nkw / D * D;
            for (int base = 0; base < nmain; base += D) {
#pragma unroll
                for (int uu = 0; uu < D; ++uu) {
                    loadB((uu + 1) & 1, base + uu + 1);
                    fn_wait_vm<NLA * (D - 1)>();
                    mma(uu, uu & 1);
                    loadA(uu, base + uu + D);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            fn_wait_vm<0>();
#pragma unroll
            for (int uu = 0; uu < D; ++uu)
                if (nmain + uu < nkw) {
                    loadB((uu + 1) & 1, nmain + uu + 1);
                    mma(uu, uu & 1);
                }
#pragma unroll
            for (int s = 0; s < D; ++s)
#pragma unroll
                for (int m = 0; m < MT; ++m) { fn_keep(fa[s][m][0]); fn_keep(fa[s][m][1]); }
        }

        FN_PSTAMP(3);
        // (d) accumulators -> LDS (padded MFMA C layout)
        if (!hand_now) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
                *reinterpret_cast<f32x4*>(red + (long)(wk * EM + wm * MT + m) * RT + lane * 4 + (lane >> 4) * 4) = acc[m][0] + acc[m][1];
        }
        __syncthreads();

        FN_PSTAMP(4);
        // (e) gate backward, element-wise
        const bool publish = q > 0 || (q == 0 && S.dh0 != nullptr);
        float* xout = S.xf + (long)(it & 1) * FS3;
        const int nkc = nk3;
        f32x4 o_r[NI], o_z[NI], o_n[NI], o_g[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            f32x4 dh;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WK; ++w) {
                    float x = red[(long)(w * EM) * RT + icoff[i] + c * 4];
                    if (hand_now) x += red[(long)((WK + w) * EM) * RT + icoff[i] + c * 4];      // even-k + odd-k accumulator, as acc[m][0] + acc[m][1]
                    a += x;
                }
                dh[c] = (a + carry[i][c]) + ext[i][c];
            }
            if (q < 0) {
                if (iact[i]) stv4(S.dh0 + (long)ib[i] * H + jj0, dh);
                continue;
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float dr, dz, dnp, dnr, cy;
                fn_gru_gate_bwd(dh[c], g_r[i][c], g_z[i][c], g_n[i][c], g_hn[i][c], hpv[i][c], dr, dz, dnp, dnr, cy);
                o_r[i][c] = dr; o_z[i][c] = dz; o_n[i][c] = dnp; o_g[i][c] = dnr; carry[i][c] = cy;
            }
            if (publish && iact[i]) {
                stv4_sc1(xout + frag_off(ib[i], jj0, nkc), o_r[i]);
                stv4_sc1(xout + frag_off(ib[i], H + jj0, nkc), o_z[i]);
                stv4_sc1(xout + frag_off(ib[i], 2 * H + jj0, nkc), o_g[i]);
            }
        }

        FN_PSTAMP(5);
        // (f) publish
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        FN_PSTAMP(6);
        // (g) outputs nobody in this launch waits for
        if (q >= 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                rs[i][0] += o_r[i]; rs[i][1] += o_z[i]; rs[i][2] += o_n[i]; rsn[i] += o_g[i];
                if (!iact[i]) continue;
                float* dg = S.dgx_all + (long)q * 3 * BH + (long)ib[i] * 3 * H + jj0;
                stv4(dg, o_r[i]);
                stv4(dg + H, o_z[i]);
                stv4(dg + 2 * H, o_n[i]);
                stv4(S.dghn_all + (long)q * BH + (long)ib[i] * H + jj0, o_g[i]);
            }
        }
        FN_PSTAMP(7);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (!iact[i]) continue;
        if (S.rowsum) {
            float* p = S.rowsum + (long)ib[i] * 3 * H + jj0;
#pragma unroll
            for (int q = 0; q < 3; ++q) stv4(p + q * H, ldv4(p + q * H) + rs[i][q]);
        }
        if (S.rowsum_n) {
            float* p = S.rowsum_n + (long)ib[i] * H + jj0;
            stv4(p, ldv4(p) + rsn[i]);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Ping-pong forms of the two scans above (round 4; kloop2_asm.h has the why and the protocol): a workgroup owns TWO halves of its row
// group - every wave one 16-row tile of each - and alternates between them.  Phase f = (half X = f & 1, step p = f >> 1): K loop of X
// (the operand ring was requested a whole epilogue ago) -> accumulators to LDS -> gate epilogue of X.  The arrival for an epilogue's
// stores is made inside the NEXT phase's K loop, the counter of the other half is read there as well, the other half's ring is requested
// at its end.  H = 512, full row groups (every lane of every wave issues every store: the K loops count them), saved gates required.
// Results are bit-identical to the kernels above (same accumulation order, same epilogue arithmetic).
// ---------------------------------------------------------------------------------------------------------------
FN_DEVINL void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// plain (write-back) 16-byte store as an asm statement: the ping-pong K loops count the stores of an epilogue
FN_DEVINL void stv4_asm(float* p, const f32x4& v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }

// waits (this wave) until *c >= target; false = gave up (sticky error word raised, `dead` set: every wave leaves behind the next barrier)
FN_DEVINL int lds_ld(unsigned a) {
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
    return v;
}
FN_DEVINL void lds_st(unsigned a, int v) { asm volatile("ds_write_b32 %0, %1" ::"v"(a), "v"(v) : "memory"); }
FN_DEVINL bool pp_wait_counter(u32* c, u32 target, u32* err, unsigned dead) {
    u32 spins = 0;
    while (ld_cnt(c) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0 && (spins > SPIN_LIMIT || ld_cnt(err) != 0)) {
            __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            lds_st(dead, 1);
            return false;
        }
    }
    return true;
}

template <int WK>
__global__ __launch_bounds__(NT) void gru_fwd_pp_kernel(const PArgs args) {
    static_assert(WK == 1 || WK == 2, "128-row groups (one wave over all of K) or 64-row groups (K split in two)");
    constexpr int WM = 4 / WK, EM = 2 * WM;          // wave rows; row tiles per workgroup (one per wave row and half)
    constexpr int H = 512, nk = 16, nslices = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                // [3][nk][2][64][4]  W_hh slice, B-fragment order
    float* red = smem + 3 * H * 16;                  // [WK][EM][3][RT]    accumulator exchange
    // "a wave gave up" flag: an LDS word behind the accumulator tiles, accessed with ds_read / ds_write through its LDS address (as a C++
    // object it is reached through FLAT instructions, which count on vmcnt: the compiler then drains the ring in flight)
    const unsigned dead = lds_addr(red + WK * EM * 3 * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const PScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (16 * EM), hh0 = slice * 16;
    const int nrt = B >> 4;
    const long FS = (long)nrt * 16 * H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wk = wave / WM;
    u32* cnt[2] = {args.sync + (2 * g) * 32, args.sync + (2 * g + 1) * 32};      // one arrival counter per half
    u32* err = args.err;

    if (tid == 0) lds_st(dead, 0);
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const float4* src = reinterpret_cast<const float4*>(S.w_frag + (long)(q * nslices + slice) * nk * 512);
        float4* dst = reinterpret_cast<float4*>(wl + (long)q * nk * 512);
        for (int i = tid; i < nk * 128; i += NT) dst[i] = src[i];
    }

    // epilogue item of this thread inside a half: (row, 4 consecutive units); 64-row groups have 128 items per half: lanes 0-31 of every
    // wave (no wave without an item: every wave issues every store instruction)
    const bool has_item = WK == 1 || lane < 32;
    const int item = WK == 1 ? tid : wave * 32 + (lane & 31);
    const int rl = item >> 2, th = rl >> 4;
    const int jj0 = hh0 + 4 * (item & 3);
    int ib[2], icoff[2];
    f32x4 bh[3], bi[3], e_rb[2][3], hp[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int tile = 2 * th + hx;
        ib[hx] = m0 + tile * 16 + (rl & 15);
        icoff[hx] = (tile * 3) * RT + ((rl & 15) >> 2) * 68 + (item & 3) * 16 + (rl & 3);
        hp[hx] = S.h0 ? ldv4(S.h0 + (long)ib[hx] * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            e_rb[hx][q] = S.gx_rowbias ? ldv4(S.gx_rowbias + (long)ib[hx] * 3 * H + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bh[q] = ldv4(S.b_hh + q * H + jj0);
        bi[q] = S.b_ih ? ldv4(S.b_ih + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int tokn[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int tau0 = (S.reverse ? T - 1 : 0) + S.idx_shift;
        tokn[hx] = (S.gx_table && tau0 >= 0) ? S.idx[(long)ib[hx] * S.idx_ld + tau0] : S.start_token;
    }
    __syncthreads();

    const int c0 = nk * wk / WK;                     // this wave's first K chunk
    unsigned vo[2], h_red[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int tw = 2 * wm + hx;                  // this wave's row tile of half hx
        vo[hx] = (unsigned)((((long)(m0 >> 4) + tw) * nk * 512 + lane * 4) * 4);
        h_red[hx] = lds_addr(red) + ((((wk * EM + tw) * 3) * RT + lane * 4 + (lane >> 4) * 4) * 4);
    }
    const unsigned h_lp = lds_addr(wl) + c0 * 2048 + lane * 16;
    const int* legal_i = S.idx ? S.idx : reinterpret_cast<const int*>(S.b_hh);

    // stores of the last epilogue: issued by the NEXT phase's K loop statement (kloop2_asm.h), or by flush_stores() when none follows
    float *st_a0 = nullptr, *st_a1 = nullptr, *st_a2 = nullptr;
    f32x4 st_d[5];
    auto flush_stores = [&]() __attribute__((always_inline)) {
        if (has_item) {
            stv4_sc1(st_a0, st_d[0]);
            stv4(st_a1, st_d[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) stv4(st_a2 + q * 256, st_d[1 + q]);
        }
    };
    // gate epilogue of half hx at step p: reads the accumulator tiles from LDS, leaves the new state (exchange slab + h_all) and the saved
    // gates in st_*: arithmetic and LDS reads only, no vector-memory instruction
    auto epilogue = [&](auto HX, const int p, const f32x4 (&e_x)[3]) __attribute__((always_inline)) {
        constexpr int hx = decltype(HX)::value;
        float* xout = (p + 1 < T || !S.hlf) ? S.xf + (long)((p + 1) & 1) * FS : S.hlf;      // last step: the next launch's hand-over slab (or a slab nobody reads)
        f32x4 gh[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WK; ++w) a += red[(long)(w * EM * 3 + q) * RT + icoff[hx] + c * 4];
                gh[q][c] = a + bh[q][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r, z, n, hn;
            fn_gru_gate((bi[0][c] + e_x[0][c]) + e_rb[hx][0][c], (bi[1][c] + e_x[1][c]) + e_rb[hx][1][c], (bi[2][c] + e_x[2][c]) + e_rb[hx][2][c],
                        gh[0][c], gh[1][c], gh[2][c], hp[hx][c], r, z, n, hn);
            hp[hx][c] = hn;
            st_d[1][c] = r; st_d[2][c] = z; st_d[3][c] = n; st_d[4][c] = gh[2][c];
        }
        st_d[0] = hp[hx];
        st_a0 = xout + frag_off(ib[hx], jj0, nk);
        st_a1 = S.h_all + (long)p * B * H + (long)ib[hx] * H + jj0;
        st_a2 = S.gates + (long)p * 4 * H * nrt * 16 + gate_off(ib[hx], 0, jj0, nrt);
    };
    auto next_token = [&](const int hx, const int p, const int loaded) {      // token of step p + 1 (loaded = idx value requested for it)
        const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
        return tau1 >= 0 ? loaded : S.start_token;
    };
    auto token_addr = [&](const int hx, const int p) {                        // where the token of step p + 1 lives (a legal address otherwise)
        const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
        return (S.gx_table && p + 1 < T && tau1 >= 0) ? S.idx + (long)ib[hx] * S.idx_ld + tau1 : legal_i;
    };

    int pend = -1;                                   // half whose epilogue stores still await their arrival (made inside the next K loop)
    int p_first = 0;
    if (!S.h0) {
        // ---- step 0 from a zero state: no K loop (gh = b_hh); half A arrives at once, half B's arrival rides in the first K loop --------
        p_first = 1;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
#pragma unroll
            for (int n = 0; n < 3; ++n)
                *reinterpret_cast<f32x4*>(red + (long)((wk * EM + 2 * wm + hx) * 3 + n) * RT + lane * 4 + (lane >> 4) * 4) = z4;
        f32x4 e0[2][3];
        int t1[2];
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                e0[hx][q] = z4;
                if (S.gx_table) e0[hx][q] = ldv4(S.gx_table + (long)tokn[hx] * 3 * H + jj0 + q * H);
                if (S.gx_dense) e0[hx][q] += ldv4(S.gx_dense + (long)ib[hx] * 3 * H + jj0 + q * H);
            }
            t1[hx] = *token_addr(hx, 0);
        }
        __syncthreads();
        epilogue(std::integral_constant<int, 0>{}, 0, e0[0]);
        flush_stores();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        epilogue(std::integral_constant<int, 1>{}, 0, e0[1]);
        flush_stores();
        pend = 1;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
            if (S.gx_table) tokn[hx] = next_token(hx, 0, t1[hx]);
    }
    // every load the compiler knows about has to be complete HERE: a value still pending at the loop entry would make it place an
    // s_waitcnt vmcnt(0) at its first use inside the loop - executed by every iteration, draining the rings in flight
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        asm volatile("" : "+v"(tokn[hx]));
        fn_touch(hp[hx]);
#pragma unroll
        for (int q = 0; q < 3; ++q) fn_touch(e_rb[hx][q]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { fn_touch(bh[q]); fn_touch(bi[q]); }

    // ---- ring of the first K phase (half A of step p_first) ---------------------------------------------------------------------------
    {
        const float* xin = ((p_first == 0 && S.h0f) ? S.h0f : S.xf + (long)(p_first & 1) * FS) + (long)c0 * 512;
        if (p_first > 0) pp_wait_counter(cnt[0], (u32)nslices * (u32)p_first, err, dead);
        if (WK == 1) fn_pp_fwd_k512_pro(xin, vo[0], h_lp, h_lp + 65536u);
        else fn_pp_fwd_k256_pro(xin, vo[0], h_lp, h_lp + 65536u);
    }
    bool first = true;                               // no epilogue stores are waiting to be issued (first K phase of the launch)

    auto phase = [&](auto HX, const int p) __attribute__((always_inline)) -> bool {
        constexpr int hx = decltype(HX)::value, hy = hx ^ 1;
        const int f1 = 2 * p + hx + 1, p1 = f1 >> 1;
        const bool next_k = f1 < 2 * T;
        const float* xin = ((p == 0 && S.h0f) ? S.h0f : S.xf + (long)(p & 1) * FS) + (long)c0 * 512;
        const float* xiny = S.xf + (long)(p1 & 1) * FS + (long)c0 * 512;      // p1 >= 1 whenever a next phase exists beyond (A, 0) -> (B, 0)
        if (p1 == 0 && S.h0f) xiny = S.h0f + (long)c0 * 512;
        const unsigned ptgt = next_k ? (u32)nslices * (u32)p1 : 0xffffffffu;
        const float* xa = S.gx_table ? S.gx_table + (long)tokn[hx] * 3 * H + jj0 + H : S.gx_dense + ((long)p * B + ib[hx]) * 3 * H + jj0 + H;
        const int* ta = token_addr(hx, p);
        f32x4 e_x[3];
        int tk;
        unsigned pv;
        const int arr = pend >= 0 ? (wave == 0 ? 2 : 1) : 0;
        u32* acnt = cnt[pend > 0 ? 1 : 0];
        FN_PSTAMP(hx * 4 + 0);
        const unsigned h_lq = h_lp + 65536u;
        if (WK == 1) {
            if (first) fn_pp_fwd_k512_first(xin, vo[hx], h_lp, h_lq, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], xa, ta, e_x, tk, pv);
            else fn_pp_fwd_k512_main(xin, vo[hx], h_lp, h_lq, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], xa, ta, st_a0, st_a1, st_a2, st_d[0], st_d[1], st_d[2],
                                     st_d[3], st_d[4], e_x, tk, pv);
        } else {
            if (first) fn_pp_fwd_k256_first(xin, vo[hx], h_lp, h_lq, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], xa, ta, e_x, tk, pv);
            else fn_pp_fwd_k256_main(xin, vo[hx], h_lp, h_lq, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], xa, ta, st_a0, st_a1, st_a2, st_d[0], st_d[1], st_d[2],
                                     st_d[3], st_d[4], e_x, tk, pv);
        }
        first = false;
        FN_PSTAMP(hx * 4 + 1);
        if (S.gx_table) tokn[hx] = next_token(hx, p, tk);
        if (next_k && pv < ptgt) {                   // rare: the other half's inputs were not all published yet - poll, then request its ring
            FN_PCOUNT(0);
            pp_wait_counter(cnt[hy], ptgt, err, dead);
            if (WK == 1) fn_pp_fwd_k512_pro(xiny, vo[hy], h_lp, h_lp + 65536u);
            else fn_pp_fwd_k256_pro(xiny, vo[hy], h_lp, h_lp + 65536u);
        }
        lds_barrier();
        FN_PSTAMP(hx * 4 + 2);
        if (lds_ld(dead)) return false;
        epilogue(HX, p, e_x);
        FN_PSTAMP(hx * 4 + 3);
        pend = p + 1 < T ? hx : -1;                  // the arrival is made inside the next phase's K loop
        return true;
    };

#pragma unroll 1
    for (int p = p_first; p < T; ++p) {
        if (!phase(std::integral_constant<int, 0>{}, p)) return;
        if (!phase(std::integral_constant<int, 1>{}, p)) return;
    }
    flush_stores();                                  // the last epilogue's (no K loop follows)
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong forward scan on the bf16 MFMA with exact bf16 triple splits (round 5; kloop3_asm.h / gen_kloop3.py have the protocol): the structure of
// gru_fwd_pp_kernel - two halves per workgroup, every wave one 16-row tile of each, stores / arrival / counter read of a phase inside the next
// phase's K loop - with the operands of gru_fwd_x6_kernel: the W_hh slice as bf16 triples in LDS (144 KB), the recurrent state exchanged as
// triples (rounded pieces, fn_split3x4), 18 v_mfma_f32_16x16x32_bf16 per (row tile, 32 k).  The accumulator tiles in LDS hold ONE half (13 KB):
// the s_barrier every K loop executes at its arrival point separates a phase's accumulator writes from the previous epilogue's reads.
// The ring of the next phase is requested by its own statement behind the K loop (fn_x6_fwd_*_pro) once the counter value the loop read
// near its end says that the other half's inputs are all there (else: poll first).  WK = 1: 128-row groups; WK = 2: 64-row groups, K split
// over two wave pairs (not the summation order of gru_fwd_x6_kernel's 64-row form, which has no K split).
// H = 512, full row groups, saved gates, exactly one of gx_table / gx_dense, T >= 2.
// ---------------------------------------------------------------------------------------------------------------
template <int WK>
__global__ __launch_bounds__(NT) void gru_fwd_x6pp_kernel(const PArgs args) {
    static_assert(WK == 1 || WK == 2, "128-row groups (one wave over all of K) or 64-row groups (K split in two)");
    constexpr int WM = 4 / WK, EM = 2 * WM;          // wave rows; row tiles per workgroup (one per wave row and half)
    constexpr int H = 512, NB = 16, nslices = 32;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);      // [3][NB][3][64] weight triples, B-operand order
    float* red = smem + 3 * NB * 3 * 64 * 4;         // [WK][WM][3][RT] accumulator exchange of the CURRENT half
    const unsigned dead = lds_addr(red + WK * WM * 3 * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const PScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (16 * EM), hh0 = slice * 16;
    const int nrt = B >> 4;
    const long FSB = (long)nrt * NB * 3 * 1024;      // bytes of one exchange slab
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave % WM, wk = wave / WM;
    u32* cnt[2] = {args.sync + (2 * g) * 32, args.sync + (2 * g + 1) * 32};      // one arrival counter per half
    u32* err = args.err;
    char* xs = reinterpret_cast<char*>(S.xf);
    const char* h0f = reinterpret_cast<const char*>(S.h0f);

    if (tid == 0) lds_st(dead, 0);
#pragma unroll 1
    for (int q = 0; q < 3; ++q) {
        const u32x4* src = reinterpret_cast<const u32x4*>(S.w_frag) + (long)(q * nslices + slice) * NB * 3 * 64;
        u32x4* dst = wl + q * NB * 3 * 64;
        for (int i = tid; i < NB * 3 * 64; i += NT) dst[i] = src[i];
    }

    // epilogue item of this thread inside a half: (row, 4 consecutive units); 64-row groups have 128 items per half: lanes 0-31 of every wave
    const bool has_item = WK == 1 || lane < 32;
    const int item = WK == 1 ? tid : wave * 32 + (lane & 31);
    const int rl = item >> 2, th = rl >> 4, u4 = item & 3;
    const int jj0 = hh0 + 4 * u4;
    int ib[2];
    long soff[2];
    f32x4 bh[3], bi[3], e_rb[2][3], hp[2];
    const int icoff = (th * 3) * RT + ((rl & 15) >> 2) * 68 + u4 * 16 + (rl & 3);      // tile th of the current half
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int tile = 2 * th + hx;
        ib[hx] = m0 + tile * 16 + (rl & 15);
        // where this item's four new state values go on the exchange slab: row tile, k block = slice / 2, k group 2 (slice & 1) + u4 / 2, half u4 & 1
        soff[hx] = ((((long)(ib[hx] >> 4) * NB + (slice >> 1)) * 3) * 64 + (ib[hx] & 15) + 16 * (2 * (slice & 1) + (u4 >> 1))) * 16 + (u4 & 1) * 8;
        hp[hx] = S.h0 ? ldv4(S.h0 + (long)ib[hx] * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q)
            e_rb[hx][q] = S.gx_rowbias ? ldv4(S.gx_rowbias + (long)ib[hx] * 3 * H + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        bh[q] = ldv4(S.b_hh + q * H + jj0);
        bi[q] = S.b_ih ? ldv4(S.b_ih + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    int tokn[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int tau0 = (S.reverse ? T - 1 : 0) + S.idx_shift;
        tokn[hx] = (S.gx_table && tau0 >= 0) ? S.idx[(long)ib[hx] * S.idx_ld + tau0] : S.start_token;
    }
    __syncthreads();

    const int c0 = NB * wk / WK;                     // this wave's first K block
    unsigned vo[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) vo[hx] = (unsigned)((((long)(m0 >> 4) + 2 * wm + hx) * NB * 3 * 1024) + lane * 16);
    const unsigned h_red = lds_addr(red) + ((((wk * WM + wm) * 3) * RT + lane * 4 + (lane >> 4) * 4) * 4);
    const unsigned lp0 = lds_addr(wl) + c0 * 3072 + lane * 16, lp1 = lp0 + 49152u, lp2 = lp0 + 98304u;
    const int* legal_i = S.idx ? S.idx : reinterpret_cast<const int*>(S.b_hh);

    // stores of the last epilogue: issued by the NEXT phase's K loop statement (kloop3_asm.h), or by flush_stores() when none follows
    char* st_a0 = nullptr;
    float *st_a1 = nullptr, *st_a2 = nullptr;
    u32x2 st_t[3];
    f32x4 st_d[5];
    auto flush_stores = [&]() __attribute__((always_inline)) {
        if (has_item) {
            stv2_sc1(st_a0, st_t[0]);
            stv2_sc1(st_a0 + 1024, st_t[1]);
            stv2_sc1(st_a0 + 2048, st_t[2]);
            stv4(st_a1, st_d[0]);
#pragma unroll
            for (int q = 0; q < 4; ++q) stv4(st_a2 + q * 256, st_d[1 + q]);
        }
    };
    // gate epilogue of half hx at step p: reads the accumulator tiles from LDS, leaves the new state (triples for the exchange slab + h_all) and the
    // saved gates in st_*: arithmetic and LDS reads only, no vector-memory instruction
    auto epilogue = [&](auto HX, const int p, const f32x4 (&e_x)[3]) __attribute__((always_inline)) {
        constexpr int hx = decltype(HX)::value;
        char* xout = (p + 1 < T || !S.hlf) ? xs + (long)((p + 1) & 1) * FSB : reinterpret_cast<char*>(S.hlf);      // last step: the next launch's hand-over image (or a slab nobody reads)
        f32x4 gh[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float a = 0.f;
#pragma unroll
                for (int w = 0; w < WK; ++w) a += red[(long)(w * WM * 3 + q) * RT + icoff + c * 4];
                gh[q][c] = a + bh[q][c];
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float r, z, n, hn;
            fn_gru_gate((bi[0][c] + e_x[0][c]) + e_rb[hx][0][c], (bi[1][c] + e_x[1][c]) + e_rb[hx][1][c], (bi[2][c] + e_x[2][c]) + e_rb[hx][2][c],
                        gh[0][c], gh[1][c], gh[2][c], hp[hx][c], r, z, n, hn);
            hp[hx][c] = hn;
            st_d[1][c] = r; st_d[2][c] = z; st_d[3][c] = n; st_d[4][c] = gh[2][c];
        }
        st_d[0] = hp[hx];
        fn_split3x4(hp[hx], st_t[0], st_t[1], st_t[2]);
        st_a0 = xout + soff[hx];
        st_a1 = S.h_all + (long)p * B * H + (long)ib[hx] * H + jj0;
        st_a2 = S.gates + (long)p * 4 * H * nrt * 16 + gate_off(ib[hx], 0, jj0, nrt);
    };
    auto next_token = [&](const int hx, const int p, const int loaded) {      // token of step p + 1 (loaded = idx value requested for it)
        const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
        return tau1 >= 0 ? loaded : S.start_token;
    };
    auto token_addr = [&](const int hx, const int p) {                        // where the token of step p + 1 lives (a legal address otherwise)
        const int tau1 = (S.reverse ? T - 2 - p : p + 1) + S.idx_shift;
        return (S.gx_table && p + 1 < T && tau1 >= 0) ? S.idx + (long)ib[hx] * S.idx_ld + tau1 : legal_i;
    };
    auto request = [&](const char* xin, unsigned v) __attribute__((always_inline)) {
        if (WK == 1) fn_x6_fwd_k512_pro(xin, v, lp0, lp1, lp2);
        else fn_x6_fwd_k256_pro(xin, v, lp0, lp1, lp2);
    };

    int pend = -1;                                   // half whose epilogue stores still await their arrival (made inside the next K loop)
    int p_first = 0;
    if (!S.h0) {
        // ---- step 0 from a zero state: no K loop (gh = b_hh); half A arrives at once, half B's arrival rides in the first K loop --------
        p_first = 1;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int n = 0; n < 3; ++n)
            *reinterpret_cast<f32x4*>(red + (long)((wk * WM + wm) * 3 + n) * RT + lane * 4 + (lane >> 4) * 4) = z4;
        f32x4 e0[2][3];
        int t1[2];
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                e0[hx][q] = z4;
                if (S.gx_table) e0[hx][q] = ldv4(S.gx_table + (long)tokn[hx] * 3 * H + jj0 + q * H);
                if (S.gx_dense) e0[hx][q] += ldv4(S.gx_dense + (long)ib[hx] * 3 * H + jj0 + q * H);
            }
            t1[hx] = *token_addr(hx, 0);
        }
        __syncthreads();
        epilogue(std::integral_constant<int, 0>{}, 0, e0[0]);
        flush_stores();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        epilogue(std::integral_constant<int, 1>{}, 0, e0[1]);
        flush_stores();
        pend = 1;
#pragma unroll
        for (int hx = 0; hx < 2; ++hx)
            if (S.gx_table) tokn[hx] = next_token(hx, 0, t1[hx]);
    }
    // every load the compiler knows about has to be complete HERE (see gru_fwd_pp_kernel)
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        asm volatile("" : "+v"(tokn[hx]));
        fn_touch(hp[hx]);
#pragma unroll
        for (int q = 0; q < 3; ++q) fn_touch(e_rb[hx][q]);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { fn_touch(bh[q]); fn_touch(bi[q]); }

    auto slab_in = [&](const int p) { return ((p == 0 && h0f) ? h0f : xs + (long)(p & 1) * FSB) + (long)c0 * 3072; };
    // ---- ring of the first K phase (half A of step p_first) ---------------------------------------------------------------------------
    if (p_first > 0) pp_wait_counter(cnt[0], (u32)nslices * (u32)p_first, err, dead);
    request(slab_in(p_first), vo[0]);
    bool first = true;                               // no epilogue stores are waiting to be issued (first K phase of the launch)

    auto phase = [&](auto HX, const int p) __attribute__((always_inline)) -> bool {
        constexpr int hx = decltype(HX)::value, hy = hx ^ 1;
        const int f1 = 2 * p + hx + 1, p1 = f1 >> 1;
        const bool next_k = f1 < 2 * T;
        const unsigned ptgt = (u32)nslices * (u32)p1;
        const float* xa = S.gx_table ? S.gx_table + (long)tokn[hx] * 3 * H + jj0 + H : S.gx_dense + ((long)p * B + ib[hx]) * 3 * H + jj0 + H;
        const int* ta = token_addr(hx, p);
        f32x4 e_x[3];
        int tk;
        unsigned pv;
        const int arr = pend >= 0 ? (wave == 0 ? 2 : 1) : 0;
        u32* acnt = cnt[pend > 0 ? 1 : 0];
        FN_PSTAMP(hx * 4 + 0);
        if (WK == 1) {
            if (first) fn_x6_fwd_k512_first(slab_in(p), vo[hx], lp0, lp1, lp2, h_red, arr, acnt, cnt[hy], xa, ta, e_x, tk, pv);
            else fn_x6_fwd_k512_main(slab_in(p), vo[hx], lp0, lp1, lp2, h_red, arr, acnt, cnt[hy], xa, ta, st_a0, st_a1, st_a2, st_t[0], st_t[1], st_t[2],
                                     st_d[0], st_d[1], st_d[2], st_d[3], st_d[4], e_x, tk, pv);
        } else {
            if (first) fn_x6_fwd_k256_first(slab_in(p), vo[hx], lp0, lp1, lp2, h_red, arr, acnt, cnt[hy], xa, ta, e_x, tk, pv);
            else fn_x6_fwd_k256_main(slab_in(p), vo[hx], lp0, lp1, lp2, h_red, arr, acnt, cnt[hy], xa, ta, st_a0, st_a1, st_a2, st_t[0], st_t[1], st_t[2],
                                     st_d[0], st_d[1], st_d[2], st_d[3], st_d[4], e_x, tk, pv);
        }
        first = false;
        FN_PSTAMP(hx * 4 + 1);
        if (S.gx_table) tokn[hx] = next_token(hx, p, tk);
        if (next_k) {
            // the other half's inputs: all published when the loop looked (the usual case), else poll; then its ring, which lands under the epilogue
            if (p1 > 0 && __builtin_amdgcn_readfirstlane((int)pv) < (int)ptgt) {
                FN_PCOUNT(0);
                pp_wait_counter(cnt[hy], ptgt, err, dead);
            }
            request(slab_in(p1), vo[hy]);
        }
        lds_barrier();
        FN_PSTAMP(hx * 4 + 2);
        if (lds_ld(dead)) return false;
        epilogue(HX, p, e_x);
        FN_PSTAMP(hx * 4 + 3);
        pend = p + 1 < T ? hx : -1;                  // the arrival is made inside the next phase's K loop
        return true;
    };

#pragma unroll 1
    for (int p = p_first; p < T; ++p) {
        if (!phase(std::integral_constant<int, 0>{}, p)) return;
        if (!phase(std::integral_constant<int, 1>{}, p)) return;
    }
    flush_stores();                                  // the last epilogue's (no K loop follows)
}

// backward scan, ping-pong form with HALF of the W_hh^T slice register-stationary (kloop2_asm.h, GenRS): a workgroup owns 32 dh columns
// (16 slices) of a 32 TH-row group; column tile 0 of its slice lives in AGPRs (every wave its K quarter), column tile 1 in LDS; every wave
// multiplies all TH row tiles of the current half over its K quarter, the epilogue adds the four partial sums in wave order.  One operand
// load feeds 8 MFMAs (4 in the kernels above), the operand stream per workgroup and the weight-fragment LDS reads halve.
// NOT bit-identical to the kernels above (another summation order over K).
template <int TH>
__global__ __launch_bounds__(NT) void gru_bwd_rs_kernel(const QArgs args) {
    // The kernel claims the WHOLE register file (512 registers per lane; it needs 372): no wave of another kernel can sit on a SIMD beside a scan
    // wave.  With a co-resident wave (the <= 128-register projection GEMM of the aux lane) one workgroup of a <1> launch came out with a perturbed
    // 16 x 32 patch about once in 2500 eager steps (profiles/r05_eager_nondeterminism.txt: cause not found, narrowed down to "a foreign wave on the
    // same SIMD"; never seen with this claim: profiles/r06_whole_rf_soak.txt).  Cost: the co-running GEMM runs behind the scan instead of beside it,
    // < 0.1 % of the step.
    asm volatile("" ::: "v255", "a255");
    static_assert(TH == 1 || TH == 2, "row tiles per half: 32-row or 64-row groups");
    constexpr int TT = 2 * TH;                       // accumulator tiles per wave and half
    constexpr int H = 512, nk3 = 48, nslices = 16;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                                // [nk3][2][64][4]  column tile 1 of the W_hh^T slice, B-fragment order
    float* red = smem + 3 * H * 16;                  // [2 halves][4 waves][TT][RT]
    const unsigned dead = lds_addr(red + 2 * 4 * TT * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
#ifdef FN_PIN_ARGS
    // experiment (round 5): every field of the scan descriptor read from the kernarg segment ONCE and pinned in SGPRs (the compiler otherwise re-loads
    // them from the kernarg segment wherever it is short of SGPRs - 40 scalar loads inside the phase loop)
    QScan S = args.s[si];
    {
        auto pin = [](auto& p) __attribute__((always_inline)) { asm volatile("" : "+s"(p)); };
        pin(S.wt_frag); pin(S.h0); pin(S.h_all); pin(S.gates); pin(S.dh_last); pin(S.dh_ext); pin(S.dgx_all); pin(S.dghn_all); pin(S.dh0);
        pin(S.rowsum); pin(S.rowsum_n); pin(S.xf); pin(S.B); pin(S.T); pin(S.group0);
    }
#else
    const QScan& S = args.s[si];
#endif
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (32 * TH), hh0 = slice * 32;
    const int nrt = B >> 4;
    const long FS3 = (long)nrt * 16 * 3 * H;
    const long BH = (long)B * H;
    const long GS = (long)4 * H * nrt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* cnt[2] = {args.sync + (2 * g) * 32, args.sync + (2 * g + 1) * 32};
    u32* err = args.err;

    if (tid == 0) lds_st(dead, 0);
    {
        const float4* src = reinterpret_cast<const float4*>(S.wt_frag + (long)(2 * slice + 1) * nk3 * 512);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < nk3 * 128; i += NT) dst[i] = src[i];
    }
    // column tile 0: this wave's K quarter (chunks [12 wave, 12 wave + 12)) -> AGPRs, once
    if (TH == 2) fn_rs_bwd_t2_wload(S.wt_frag + (long)(2 * slice) * nk3 * 512 + (long)wave * 12 * 512, (unsigned)lane * 16u);
    else fn_rs_bwd_t1_wload(S.wt_frag + (long)(2 * slice) * nk3 * 512 + (long)wave * 12 * 512, (unsigned)lane * 16u);

    // epilogue item inside a half: (row, 4 consecutive dh columns of the 32); 32-row groups have 128 items per half: lanes 0-31 of every wave
    const bool has_item = TH == 2 || lane < 32;
    const int item = TH == 2 ? tid : wave * 32 + (lane & 31);
    const int rl = item >> 3, u4 = item & 7;
    const int jj0 = hh0 + 4 * u4;
    int ib[2], icoff[2];
    f32x4 carry[2], rs[2][3], rsn[2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        ib[hx] = m0 + (hx * TH + (rl >> 4)) * 16 + (rl & 15);
        icoff[hx] = (hx * 4 * TT + (rl >> 4) * 2 + (u4 >> 2)) * RT + ((rl & 15) >> 2) * 68 + (u4 & 3) * 16 + (rl & 3);
        carry[hx] = S.dh_last ? ldv4(S.dh_last + (long)ib[hx] * H + jj0) : z4;
        rsn[hx] = z4;
#pragma unroll
        for (int q = 0; q < 3; ++q) rs[hx][q] = z4;
    }
    __syncthreads();

    const int c0 = 12 * wave;                         // this wave's first K chunk
    unsigned vo[2], h_red[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        vo[hx] = (unsigned)((((long)(m0 >> 4) + hx * TH) * nk3 * 512 + lane * 4) * 4);      // first row tile of the half (the second follows nk3 * 2 KB on)
        h_red[hx] = lds_addr(red) + ((((hx * 4 + wave) * TT) * RT + lane * 4 + (lane >> 4) * 4) * 4);
    }
    const unsigned h_lp = lds_addr(wl) + c0 * 2048 + lane * 16;
    const int iters = T + (S.dh0 ? 1 : 0);

    // stores of the last epilogue: issued by the NEXT phase's K loop statement (kloop2_asm.h), or by flush_stores() when none follows
    bool st_valid = false;
    float *st_base = S.xf, *st_g = nullptr, *st_n = nullptr;
    unsigned st_o = 0;
    f32x4 st_d[4];
    auto flush_stores = [&]() __attribute__((always_inline)) {
        if (st_valid && has_item) {
            char* sb = reinterpret_cast<char*>(st_base) + st_o;
            stv4_sc1(reinterpret_cast<float*>(sb), st_d[0]);
            stv4_sc1(reinterpret_cast<float*>(sb + 0x8000), st_d[1]);
            stv4_sc1(reinterpret_cast<float*>(sb + 0x10000), st_d[3]);
            stv4(st_g - H, st_d[0]);
            stv4(st_g, st_d[1]);
            stv4(st_g + H, st_d[2]);
            stv4(st_n, st_d[3]);
        }
        st_valid = false;
    };
    // gate backward of half hx at iteration it (step q): arithmetic and LDS reads only; q < 0: only dL/dh0 is left (stored here)
    auto epilogue = [&](auto HX, const int it, const f32x4 (&gt)[4], const f32x4& hpv, const f32x4& ext, const bool hand) __attribute__((always_inline)) {
        constexpr int hx = decltype(HX)::value;
        const int q = T - 1 - it;
        f32x4 dh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
            if (hand) {
#pragma unroll
                for (int w = 0; w < 4; ++w) a += red[(long)(w * TT) * RT + icoff[hx] + c * 4];      // the four K quarters, in wave order
            }
            dh[c] = (a + carry[hx][c]) + ext[c];
        }
        if (q < 0) {
            if (has_item) stv4(S.dh0 + (long)ib[hx] * H + jj0, dh);
            st_valid = false;
            return;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float dr, dz, dnp, dnr, cy;
            fn_gru_gate_bwd(dh[c], gt[0][c], gt[1][c], gt[2][c], gt[3][c], hpv[c], dr, dz, dnp, dnr, cy);
            st_d[0][c] = dr; st_d[1][c] = dz; st_d[2][c] = dnp; st_d[3][c] = dnr; carry[hx][c] = cy;
        }
        rs[hx][0] += st_d[0]; rs[hx][1] += st_d[1]; rs[hx][2] += st_d[2]; rsn[hx] += st_d[3];
        st_base = S.xf + (long)(it & 1) * FS3;
        st_o = (unsigned)(frag_off(ib[hx], jj0, nk3) * 4);
        st_g = S.dgx_all + (long)q * 3 * BH + (long)ib[hx] * 3 * H + jj0 + H;
        st_n = S.dghn_all + (long)q * BH + (long)ib[hx] * H + jj0;
        st_valid = true;
    };

    int pend = -1;
    // ---- iteration 0: no K loop (dh = dh_last + dh_ext[T-1]); half A arrives at once, half B's arrival rides in the first K loop ----------
    {
        const int q = T - 1;
        f32x4 g0[2][4], hp0[2], x0[2];
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            const float* gq = S.gates + (long)q * GS + gate_off(ib[hx], 0, jj0, nrt);
#pragma unroll
            for (int k = 0; k < 4; ++k) g0[hx][k] = ldv4(gq + k * 256);
            hp0[hx] = q > 0 ? ldv4(S.h_all + (long)(q - 1) * BH + (long)ib[hx] * H + jj0) : (S.h0 ? ldv4(S.h0 + (long)ib[hx] * H + jj0) : z4);
            x0[hx] = S.dh_ext ? ldv4(S.dh_ext + (long)q * BH + (long)ib[hx] * H + jj0) : z4;
        }
        const bool publish = iters > 1;
        epilogue(std::integral_constant<int, 0>{}, 0, g0[0], hp0[0], x0[0], false);
        flush_stores();
        if (publish) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        epilogue(std::integral_constant<int, 1>{}, 0, g0[1], hp0[1], x0[1], false);
        flush_stores();
        pend = publish ? 1 : -1;
    }
    if (iters > 1) {
        // every load the compiler knows about has to be complete before the loop (see gru_fwd_pp_kernel)
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            fn_touch(carry[hx]);
            fn_touch(rsn[hx]);
#pragma unroll
            for (int q = 0; q < 3; ++q) fn_touch(rs[hx][q]);
        }
        {
            const float* xin = S.xf + (long)c0 * 512;          // slab 0: iteration 0's gate gradients
            pp_wait_counter(cnt[0], (u32)nslices, err, dead);
            if (TH == 2) fn_rs_bwd_t2_pro(xin, vo[0], h_lp);
            else fn_rs_bwd_t1_pro(xin, vo[0], h_lp);
        }

        auto phase = [&](auto HX, const int it) __attribute__((always_inline)) -> bool {
            constexpr int hx = decltype(HX)::value, hy = hx ^ 1;
            const int q = T - 1 - it, qc = q > 0 ? q : 0;
            const int it1 = it + hx;
            const int p = it;                             // FN_PSTAMP
            (void)p;
            const bool next_k = it1 < iters;
            const float* xin = S.xf + (long)((it - 1) & 1) * FS3 + (long)c0 * 512;
            const float* xiny = S.xf + (long)((it1 - 1) & 1) * FS3 + (long)c0 * 512;
            const unsigned ptgt = next_k ? (u32)nslices * (u32)it1 : 0xffffffffu;
            const long ro = (long)ib[hx] * H + jj0;
            const float* ga = S.gates + (long)qc * GS + gate_off(ib[hx], 0, jj0, nrt);
            const float* ha = qc > 0 ? S.h_all + (long)(qc - 1) * BH + ro : (S.h0 ? S.h0 + ro : S.h_all + ro);      // unused values still come from a legal address
            const float* xa = S.dh_ext ? S.dh_ext + (long)qc * BH + ro : S.h_all + ro;
            const bool hzero = q < 0 || (q == 0 && !S.h0), xzero = q < 0 || !S.dh_ext;
            f32x4 gt[4], hp2, xt2;
            unsigned pv;
            const int arr = pend >= 0 ? (wave == 0 ? 2 : 1) : 0;
            u32* acnt = cnt[pend > 0 ? 1 : 0];
            FN_PSTAMP(hx * 4 + 0);
            if (TH == 2) {
                if (!st_valid) fn_rs_bwd_t2_first(xin, vo[hx], h_lp, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], ga, ha, xa, gt, hp2, xt2, pv);
                else fn_rs_bwd_t2_main(xin, vo[hx], h_lp, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], ga, ha, xa, st_base, st_o, st_g, st_n, st_d[0], st_d[1], st_d[2],
                                       st_d[3], gt, hp2, xt2, pv);
            } else {
                if (!st_valid) fn_rs_bwd_t1_first(xin, vo[hx], h_lp, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], ga, ha, xa, gt, hp2, xt2, pv);
                else fn_rs_bwd_t1_main(xin, vo[hx], h_lp, h_red[hx], arr, acnt, cnt[hy], ptgt, xiny, vo[hy], ga, ha, xa, st_base, st_o, st_g, st_n, st_d[0], st_d[1], st_d[2],
                                       st_d[3], gt, hp2, xt2, pv);
            }
            st_valid = false;
            FN_PSTAMP(hx * 4 + 1);
            if (next_k && pv < ptgt) {               // rare: the other half's inputs were not all published yet - poll, then request its ring
                FN_PCOUNT(0);
                pp_wait_counter(cnt[hy], ptgt, err, dead);
                if (TH == 2) fn_rs_bwd_t2_pro(xiny, vo[hy], h_lp);
                else fn_rs_bwd_t1_pro(xiny, vo[hy], h_lp);
            }
            lds_barrier();
            FN_PSTAMP(hx * 4 + 2);
            if (lds_ld(dead)) return false;
            epilogue(HX, it, gt, hzero ? z4 : hp2, xzero ? z4 : xt2, true);
            FN_PSTAMP(hx * 4 + 3);
            pend = (q > 0 || (q == 0 && S.dh0 != nullptr)) ? hx : -1;
            return true;
        };

#pragma unroll 1
        for (int it = 1; it < iters; ++it) {
            if (!phase(std::integral_constant<int, 0>{}, it)) return;
            if (!phase(std::integral_constant<int, 1>{}, it)) return;
        }
        flush_stores();
    }
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        if (!has_item) continue;
        if (S.rowsum) {
            float* p = S.rowsum + (long)ib[hx] * 3 * H + jj0;
#pragma unroll
            for (int q = 0; q < 3; ++q) stv4(p + q * H, ldv4(p + q * H) + rs[hx][q]);
        }
        if (S.rowsum_n) {
            float* p = S.rowsum_n + (long)ib[hx] * H + jj0;
            stv4(p, ldv4(p) + rsn[hx]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Backward scan on the bf16 MFMA with exact bf16 triple splits (round 5; kloop4_asm.h / gen_kloop4.py have the protocol and the register map): the
// decomposition of gru_bwd_rs_kernel - 16 slices of 32 dh columns x groups of 32 TH rows, two halves per workgroup, every wave its K quarter of both
// column tiles - with the gate gradients exchanged as bf16 triples ([row tile][48 K blocks][piece][64 lanes][8 bf16], rounded pieces) and the
// W_hh^T slice as triples (fn_frag3_pack_t image): column tile 0 of the wave's quarter + block 11 of column tile 1 in AGPRs, blocks 0..10 of
// column tile 1 in LDS (132 KB).  Accumulator tiles in LDS: one half (the K loops' s_barrier at the arrival point is the fence).  Per phase:
//   K loop statement (exchange-slab stores of the previous epilogue, arrival, this phase's epilogue operands handed over, counter read, NEXT phase's
//   epilogue operands requested) -> counter check -> ring request of the next phase -> barrier -> gate backward (compiler code) -> dgx / dghn stores (`_out`).
// Same arithmetic per element as gru_bwd_rs_kernel (fn_gru_gate_bwd, partial sums added in wave order); the products are exact, their sums
// run in another order (six partial products per 32 k).
// ---------------------------------------------------------------------------------------------------------------
template <int TH>
__global__ __launch_bounds__(NT) void gru_bwd_x6_kernel(const QArgs args) {
    static_assert(TH == 1 || TH == 2, "row tiles per half: 32-row or 64-row groups");
    constexpr int TT = 2 * TH;                       // accumulator tiles per wave and half
    constexpr int H = 512, NB3 = 48, nslices = 16, QB = 12, LB = 11;      // K blocks per row; per wave quarter; of those in LDS
    extern __shared__ __attribute__((aligned(16))) float smem[];
    u32x4* wl = reinterpret_cast<u32x4*>(smem);      // [4 quarters][LB blocks][3 pieces][64 lanes] column tile 1, B-operand order
    float* red = smem + 4 * LB * 3 * 64 * 4;         // [4 waves][TT][RT] partial sums of the CURRENT half
    const unsigned dead = lds_addr(red + 4 * TT * RT);

    const int g = block_group(args.ngroups, args.spread), slice = block_slice(args.ngroups, args.spread);
    int si = 0;
#pragma unroll
    for (int k = 1; k < FN_MAX_SCANS; ++k)
        if (k < args.n && g >= args.s[k].group0) si = k;
    const QScan& S = args.s[si];
    const int B = S.B, T = S.T;
    const int m0 = (g - S.group0) * (32 * TH), hh0 = slice * 32;
    const int nrt = B >> 4;
    const long FSB = (long)nrt * NB3 * 3 * 1024;     // bytes of one exchange slab
    const long BH = (long)B * H;
    const long GS = (long)4 * H * nrt * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32* cnt[2] = {args.sync + (2 * g) * 32, args.sync + (2 * g + 1) * 32};
    u32* err = args.err;
    char* xs = reinterpret_cast<char*>(S.xf);
    const char* wimg = reinterpret_cast<const char*>(S.wt_frag);          // [32 column tiles][48 K blocks][3][64][16 B]

    if (tid == 0) lds_st(dead, 0);
    {
        // column tile 1 (2 slice + 1): blocks 0..10 of every K quarter -> LDS
        const u32x4* src = reinterpret_cast<const u32x4*>(wimg + (long)(2 * slice + 1) * NB3 * 3072);
        for (int i = tid; i < 4 * LB * 192; i += NT) {
            const int qd = i / (LB * 192), r = i % (LB * 192);
            wl[i] = src[(long)(qd * QB) * 192 + r];
        }
    }
    fn_x6_bwd_wload(wimg + ((long)(2 * slice) * NB3 + wave * QB) * 3072, wimg + ((long)(2 * slice + 1) * NB3 + wave * QB + LB) * 3072, (unsigned)lane * 16u);

    // epilogue item inside a half: (row, 4 consecutive dh columns of the 32); 32-row groups have 128 items per half: lanes 0-31 of every wave
    const bool has_item = TH == 2 || lane < 32;
    const int item = TH == 2 ? tid : wave * 32 + (lane & 31);
    const int rl = item >> 3, u4 = item & 7;
    const int jj0 = hh0 + 4 * u4;
    int ib[2];
    unsigned so[2];
    f32x4 carry[2], rs[2][3], rsn[2];
    const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
    const int icoff = ((rl >> 4) * 2 + (u4 >> 2)) * RT + ((rl & 15) >> 2) * 68 + (u4 & 3) * 16 + (rl & 3);
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        ib[hx] = m0 + (hx * TH + (rl >> 4)) * 16 + (rl & 15);
        // gate 0 of this item on the slab: row tile, K block = slice (32 columns per block), lane = row + 16 (u4 / 2), half u4 & 1; gates 1, 2: + 16 blocks each
        so[hx] = (unsigned)(((((long)(ib[hx] >> 4) * NB3 + slice) * 3) * 64 + (ib[hx] & 15) + 16 * (u4 >> 1)) * 16 + (u4 & 1) * 8);
        carry[hx] = S.dh_last ? ldv4(S.dh_last + (long)ib[hx] * H + jj0) : z4;
        rsn[hx] = z4;
#pragma unroll
        for (int q = 0; q < 3; ++q) rs[hx][q] = z4;
    }
    __syncthreads();

    unsigned vo[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) vo[hx] = (unsigned)((((long)(m0 >> 4) + hx * TH) * NB3 * 3072) + lane * 16);      // first row tile of the half (the second follows 144 KB on)
    const unsigned h_red = lds_addr(red) + (((wave * TT) * RT + lane * 4 + (lane >> 4) * 4) * 4);
    const unsigned lp = lds_addr(wl) + wave * LB * 3072 + lane * 16;
    const int iters = T + (S.dh0 ? 1 : 0);

    // outputs of the last epilogue: dgx / dghn leave through `_out` right behind it, the exchange-slab stores through the NEXT phase's K loop statement
    bool st_valid = false;
    char* st_base = xs;
    float *st_g = nullptr, *st_n = nullptr;
    unsigned st_o = 0;
    f32x4 st_d[4];
    u32x2 st_t[3][3];
    auto publish = [&]() __attribute__((always_inline)) {             // exchange-slab stores as ONE counted statement
        if (TH == 2) fn_x6_bwd_t2_pub(st_base, st_o, st_t[0][0], st_t[0][1], st_t[0][2], st_t[1][0], st_t[1][1], st_t[1][2], st_t[2][0], st_t[2][1], st_t[2][2]);
        else fn_x6_bwd_t1_pub(st_base, st_o, st_t[0][0], st_t[0][1], st_t[0][2], st_t[1][0], st_t[1][1], st_t[1][2], st_t[2][0], st_t[2][1], st_t[2][2]);
    };
    auto flush_outs = [&]() __attribute__((always_inline)) {
        if (st_valid && has_item) {
            stv4(st_g - H, st_d[0]);
            stv4(st_g, st_d[1]);
            stv4(st_g + H, st_d[2]);
            stv4(st_n, st_d[3]);
        }
        st_valid = false;
    };
    // gate backward of half hx at iteration it (step q): arithmetic and LDS reads only; q < 0: only dL/dh0 is left (stored here).  Returns whether
    // there is something to publish.
    auto epilogue = [&](auto HX, const int it, const f32x4 (&gt)[4], const f32x4& hpv, const f32x4& ext, const bool hand) __attribute__((always_inline)) -> bool {
        constexpr int hx = decltype(HX)::value;
        const int q = T - 1 - it;
        f32x4 dh;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = 0.f;
            if (hand) {
#pragma unroll
                for (int w = 0; w < 4; ++w) a += red[(long)(w * TT) * RT + icoff + c * 4];      // the four K quarters, in wave order
            }
            dh[c] = (a + carry[hx][c]) + ext[c];
        }
        if (q < 0) {
            if (has_item) stv4(S.dh0 + (long)ib[hx] * H + jj0, dh);
            st_valid = false;
            return false;
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float dr, dz, dnp, dnr, cy;
            fn_gru_gate_bwd(dh[c], gt[0][c], gt[1][c], gt[2][c], gt[3][c], hpv[c], dr, dz, dnp, dnr, cy);
            st_d[0][c] = dr; st_d[1][c] = dz; st_d[2][c] = dnp; st_d[3][c] = dnr; carry[hx][c] = cy;
        }
        rs[hx][0] += st_d[0]; rs[hx][1] += st_d[1]; rs[hx][2] += st_d[2]; rsn[hx] += st_d[3];
        fn_split3x4(st_d[0], st_t[0][0], st_t[0][1], st_t[0][2]);
        fn_split3x4(st_d[1], st_t[1][0], st_t[1][1], st_t[1][2]);
        fn_split3x4(st_d[3], st_t[2][0], st_t[2][1], st_t[2][2]);
        st_base = xs + (long)(it & 1) * FSB;
        st_o = so[hx];
        st_g = S.dgx_all + (long)q * 3 * BH + (long)ib[hx] * 3 * H + jj0 + H;
        st_n = S.dghn_all + (long)q * BH + (long)ib[hx] * H + jj0;
        st_valid = true;
        return true;
    };
    // addresses of the epilogue operands of (half hx, iteration it); unused values still come from a legal address
    auto ext_addr = [&](const int hx, const int it, const float*& ga, const float*& ha, const float*& xa) {
        const int q = T - 1 - it, qc = q > 0 ? q : 0;
        const long ro = (long)ib[hx] * H + jj0;
        ga = S.gates + (long)qc * GS + gate_off(ib[hx], 0, jj0, nrt);
        ha = qc > 0 ? S.h_all + (long)(qc - 1) * BH + ro : (S.h0 ? S.h0 + ro : S.h_all + ro);
        xa = S.dh_ext ? S.dh_ext + (long)qc * BH + ro : S.h_all + ro;
    };

    int pend = -1;
    // ---- iteration 0: no K loop (dh = dh_last + dh_ext[T-1]); half A arrives at once, half B's arrival rides in the first K loop ----------
    {
        const int q = T - 1;
        f32x4 g0[2][4], hp0[2], x0[2];
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            const float* gq = S.gates + (long)q * GS + gate_off(ib[hx], 0, jj0, nrt);
#pragma unroll
            for (int k = 0; k < 4; ++k) g0[hx][k] = ldv4(gq + k * 256);
            hp0[hx] = q > 0 ? ldv4(S.h_all + (long)(q - 1) * BH + (long)ib[hx] * H + jj0) : (S.h0 ? ldv4(S.h0 + (long)ib[hx] * H + jj0) : z4);
            x0[hx] = S.dh_ext ? ldv4(S.dh_ext + (long)q * BH + (long)ib[hx] * H + jj0) : z4;
        }
        const bool pubs = iters > 1;
        if (epilogue(std::integral_constant<int, 0>{}, 0, g0[0], hp0[0], x0[0], false) && pubs) publish();
        flush_outs();
        if (pubs) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (epilogue(std::integral_constant<int, 1>{}, 0, g0[1], hp0[1], x0[1], false) && pubs) publish();
        flush_outs();
        pend = pubs ? 1 : -1;
    }
    if (iters > 1) {
        // everything issued so far has to be complete before the statements start counting (half B's slab stores: its arrival in the first K loop
        // then needs no wait); every load the compiler knows about as well (see gru_fwd_pp_kernel)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int hx = 0; hx < 2; ++hx) {
            fn_touch(carry[hx]);
            fn_touch(rsn[hx]);
#pragma unroll
            for (int q = 0; q < 3; ++q) fn_touch(rs[hx][q]);
        }
        auto slab_in = [&](const int it) { return xs + (long)((it - 1) & 1) * FSB + (long)(wave * QB) * 3072; };
        auto request = [&](const char* xin, unsigned v) __attribute__((always_inline)) {
            if (TH == 2) fn_x6_bwd_t2_pro(xin, v);
            else fn_x6_bwd_t1_pro(xin, v);
        };
        f32x4 gt[4], hp2, xt2;
        {
            const float *ga, *ha, *xa;
            ext_addr(0, 1, ga, ha, xa);
            pp_wait_counter(cnt[0], (u32)nslices, err, dead);           // compiler code (its own loads and waits): nothing uncounted is in flight behind it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            fn_x6_bwd_ext(ga, ha, xa);                                 // epilogue operands of the first K phase, then its ring
            request(slab_in(1), vo[0]);
        }
        bool first = true;

        auto phase = [&](auto HX, const int it) __attribute__((always_inline)) -> bool {
            constexpr int hx = decltype(HX)::value, hy = hx ^ 1;
            const int q = T - 1 - it;
            const int it1 = it + hx;
            const int p = it;                             // FN_PSTAMP
            (void)p;
            const bool next_k = it1 < iters;
            const unsigned ptgt = (u32)nslices * (u32)it1;
            const bool hzero = q < 0 || (q == 0 && !S.h0), xzero = q < 0 || !S.dh_ext;
            const float *ga, *ha, *xa;
            ext_addr(hy, next_k ? it1 : it, ga, ha, xa);  // the NEXT phase's epilogue operands (the last phase asks for legal ones nobody reads)
            unsigned pv;
            const int arr = pend >= 0 ? (wave == 0 ? 2 : 1) : 0;
            FN_PSTAMP(hx * 4 + 0);
            if (TH == 2) {
                if (first) fn_x6_bwd_t2_first(slab_in(it), vo[hx], lp, h_red, arr, cnt[hy], ga, ha, xa, gt, hp2, xt2, pv);
                else fn_x6_bwd_t2_main(slab_in(it), vo[hx], lp, h_red, arr, cnt[hy], ga, ha, xa, st_base, st_o, st_t[0][0], st_t[0][1], st_t[0][2], st_t[1][0], st_t[1][1],
                                       st_t[1][2], st_t[2][0], st_t[2][1], st_t[2][2], gt, hp2, xt2, pv);
            } else {
                if (first) fn_x6_bwd_t1_first(slab_in(it), vo[hx], lp, h_red, arr, cnt[hy], ga, ha, xa, gt, hp2, xt2, pv);
                else fn_x6_bwd_t1_main(slab_in(it), vo[hx], lp, h_red, arr, cnt[hy], ga, ha, xa, st_base, st_o, st_t[0][0], st_t[0][1], st_t[0][2], st_t[1][0], st_t[1][1],
                                       st_t[1][2], st_t[2][0], st_t[2][1], st_t[2][2], gt, hp2, xt2, pv);
            }
            first = false;
            FN_PSTAMP(hx * 4 + 1);
            if (next_k && __builtin_amdgcn_readfirstlane((int)pv) < (int)ptgt) {      // rare: the other half's inputs were not all published yet - poll
                FN_PCOUNT(0);                                                          // (compiler code that drains vmcnt: every wait behind it only gets more patient)
                pp_wait_counter(cnt[hy], ptgt, err, dead);
            }
            // ring of the next phase: lands under the epilogue.  The last phase requests one nobody multiplies: the statements count the same operations.
            if (next_k) request(slab_in(it1), vo[hy]);
            else request(slab_in(it), vo[hx]);
            lds_barrier();
            FN_PSTAMP(hx * 4 + 2);
            if (lds_ld(dead)) return false;
            const bool pub = epilogue(HX, it, gt, hzero ? z4 : hp2, xzero ? z4 : xt2, true);
            FN_PSTAMP(hx * 4 + 3);
            const bool more = q > 0 || (q == 0 && S.dh0 != nullptr);
            pend = more ? hx : -1;
            // the exchange-slab stores of this epilogue ride in the next phase's K loop (nine per item; a phase with nothing to publish lets it store
            // its old values again, into a slab nobody reads any more: the K loop statements count nine stores); dgx / dghn leave here
            if (!pub) st_base = xs + (long)(it & 1) * FSB;
            if (TH == 2) fn_x6_bwd_t2_out(st_g, st_n, st_d[0], st_d[1], st_d[2], st_d[3]);
            else fn_x6_bwd_t1_out(st_g, st_n, st_d[0], st_d[1], st_d[2], st_d[3]);
            return true;
        };

#pragma unroll 1
        for (int it = 1; it < iters; ++it) {
            if (!phase(std::integral_constant<int, 0>{}, it)) return;
            if (!phase(std::integral_constant<int, 1>{}, it)) return;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        if (!has_item) continue;
        if (S.rowsum) {
            float* p = S.rowsum + (long)ib[hx] * 3 * H + jj0;
#pragma unroll
            for (int q = 0; q < 3; ++q) stv4(p + q * H, ldv4(p + q * H) + rs[hx][q]);
        }
        if (S.rowsum_n) {
            float* p = S.rowsum_n + (long)ib[hx] * H + jj0;
            stv4(p, ldv4(p) + rsn[hx]);
        }
    }
}

constexpr int FN_MAX_DEVICES = 32;

int cu_count() {
    static std::atomic<int> n[FN_MAX_DEVICES];       // write-once per device (a process may drive several GPUs); zero-initialised
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FN_MAX_DEVICES) return 0;
    int c = n[dev].load(std::memory_order_acquire);
    if (c == 0) {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
        c = prop.multiProcessorCount;
        n[dev].store(c, std::memory_order_release);
    }
    return c;
}

// The weight-stationary kernels spin on counters that OTHER workgroups of the same launch advance: every workgroup of the grid
// must be resident at the same time.  This is what a cooperative launch would assert; here the occupancy calculator is asked how
// many workgroups of this kernel (with its dynamic LDS) fit one CU, and a grid beyond cus x that is refused (FN_PERSIST_NA -> the
// caller takes the per-step kernels).  Kernels of OTHER streams can still delay residency; that case is covered by the bounded
// spins (sticky error word), never by a hang.
template <class Args, void (*K)(const Args)>
int launch_k(const Args& a, int grid, size_t lds, int cus, hipStream_t st) {
    auto k = K;
    // per kernel instance and device: (blocks per CU << 32 | LDS bytes asked for), packed into ONE atomic word so that a reader never
    // pairs the answer for one LDS size with another; recomputed (idempotently) when the LDS size differs
    static std::atomic<unsigned long long> cache[FN_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= FN_MAX_DEVICES) return FN_PERSIST_NA;
    unsigned long long w = cache[dev].load(std::memory_order_acquire);
    if (w == 0 || (size_t)(w & 0xffffffffull) != lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return (int)e;
        int nb = 0;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), NT, lds);
        if (e != hipSuccess) return (int)e;
        w = ((unsigned long long)(unsigned)(nb > 0 ? nb : 0x7fffffff) << 32) | (unsigned long long)lds;     // 0x7fffffff = does not fit
        cache[dev].store(w, std::memory_order_release);
    }
    const long blocks_per_cu = (w >> 32) == 0x7fffffffull ? -1 : (long)(w >> 32);
    if (blocks_per_cu < 0 || (long)grid > (long)cus * blocks_per_cu) return FN_PERSIST_NA;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

// bf16 x 6 forward scans: row tiles per wave (1 = 64-row groups, 2 = 128-row groups) of the launch, 0 = not eligible.  H = 512, every scan in
// full row groups, saved gates, T >= 2, all row groups resident at once.
int x6_row_tiles(const FnGruFwd* scans, int n_scans, int maxgroups) {
    if (scans[0].H != 512) return 0;
    long g64 = 0, g128 = 0;
    bool d64 = true, d128 = true;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruFwd& d = scans[s];
        if (d.H != 512 || !d.gates || d.T < 2) return 0;
        d64 = d64 && d.B % 64 == 0; d128 = d128 && d.B % 128 == 0;
        g64 += d.B / 64; g128 += d.B / 128;
    }
    return (d64 && g64 <= maxgroups) ? 1 : ((d128 && g128 <= maxgroups) ? 2 : 0);
}


// bf16 x 6 backward scans: row tiles per half (2 = 64-row groups, 1 = 32-row groups), 0 = not eligible.  The shapes gru_bwd_rs_kernel takes: H = 512,
// every scan in full row groups, T >= 2, more than half of the chip and at most all of it (16 slices x <= 16 groups, all resident).
int x6_bwd_tiles(const FnGruBwd* scans, int n_scans, int cus) {
    if (scans[0].H != 512) return 0;
    long g64 = 0, g32 = 0;
    bool d64 = true, d32 = true;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruBwd& d = scans[s];
        if (d.H != 512 || d.T < 2) return 0;
        g64 += d.B / 64; g32 += d.B / 32;
        d64 = d64 && d.B % 64 == 0;
        d32 = d32 && d.B % 32 == 0;
    }
    // 64-row groups (the encoder shape: 4 scans x 256 rows): 3.05 ms against 3.5 ms on the fp32 MFMA.  32-row groups (a decoder-pipeline launch: 2 scans x
    // 256 rows x 32 steps) measured SLOWER than the fp32 kernel (353 against 322-337 us: both are bound by the exchange stream through the XCD's L2, which
    // the triples make 1.5 x wider, and the 12-unit K loops are too short for the store -> arrival -> counter chain): only on request (variant bit 15)
    const bool th1_ok = (scans[0].variant & 0x8000) != 0;
    // (g64 * 16 == cus: the caller asked for exactly this with a CU budget - the half-chip launches of the decoder pipeline's backward, 8 groups on 128 CUs)
    return (d64 && (g64 > 8 || g64 * 16 == cus) && g64 <= 16 && g64 * 16 <= cus) ? 2 : (th1_ok && d32 && g32 > 8 && g32 <= 16 && g32 * 16 <= cus) ? 1 : 0;
}

}  // namespace

extern "C" size_t fn_gru_sync_ws_bytes() { return ((size_t)FN_MAX_GROUPS * 32 + 32) * 4; }

// would fn_gru_seq_fwd run this call with variant bit 14 (bf16 x 6)?  Shapes only: pointers are not looked at beyond NULL-ness of `gates`.
extern "C" int fn_gru_fwd_x6_ok(const FnGruFwd* scans, int n_scans) {
    if (!scans || n_scans < 1 || n_scans > FN_MAX_SCANS) return 0;
    int cus = cu_count();
    if (scans[0].cu_budget > 0 && scans[0].cu_budget < cus) cus = scans[0].cu_budget;
    if (cus < 32) return 0;
    const int maxgroups = cus / 32 < FN_MAX_GROUPS ? cus / 32 : FN_MAX_GROUPS;
    return x6_row_tiles(scans, n_scans, maxgroups) != 0;
}

extern "C" int fn_gru_bwd_x6_ok(const FnGruBwd* scans, int n_scans) {
    if (!scans || n_scans < 1 || n_scans > FN_MAX_SCANS) return 0;
    int cus = cu_count();
    if (scans[0].cu_budget > 0 && scans[0].cu_budget < cus) cus = scans[0].cu_budget;
    return x6_bwd_tiles(scans, n_scans, cus) != 0;
}

int fn_gru_fwd_persist(const FnGruFwd* scans, int n_scans, hipStream_t st) {
    const int H = scans[0].H;
    if (H > 512 || !scans[0].sync_ws) return FN_PERSIST_NA;
    int Tmax = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruFwd& d = scans[s];
        if (d.H != H) return FN_PERSIST_NA;
        if (d.h0_frag && !d.h0) return FN_E_NULL;              // the gate epilogue reads the row-major state
        if ((((uintptr_t)d.h0_frag) | ((uintptr_t)d.h_last_frag)) & 15) return FN_E_ALIGN;
        // the epilogue moves 16-byte vectors
        const uintptr_t al = (uintptr_t)d.b_hh | (uintptr_t)d.b_ih | (uintptr_t)d.h0 | (uintptr_t)d.gx_dense | (uintptr_t)d.gx_table |
                             (uintptr_t)d.gx_rowbias | (uintptr_t)d.h_all | (uintptr_t)d.gates;
        if (al & 15) return FN_PERSIST_NA;
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    if (Tmax < 2) return FN_PERSIST_NA;                 // nothing to keep stationary
    int cus = cu_count();
    if (scans[0].cu_budget > 0 && scans[0].cu_budget < cus) cus = scans[0].cu_budget;
    const int nslices = H / 16;
    if (cus <= 0 || nslices > cus) return FN_PERSIST_NA;
    const int maxgroups = cus / nslices < FN_MAX_GROUPS ? cus / nslices : FN_MAX_GROUPS;
    if (scans[0].variant & 0x4000) {
        // exact split products on the bf16 MFMA (gru_fwd_x6_kernel).  The caller hands over w_hh_frag = fn_frag3_pack image and
        // frag_ws of 3 * fn_frag_floats(B, H) floats.  Eligibility: x6_row_tiles() (the same predicate fn_gru_fwd_x6_ok answers with).
        for (int s = 0; s < n_scans; ++s) {
            const FnGruFwd& d = scans[s];
            if (d.h0_frag && !d.h0) return FN_E_NULL;
            if ((((uintptr_t)d.h0_frag) | ((uintptr_t)d.h_last_frag) | (uintptr_t)d.h0) & 15) return FN_E_ALIGN;
        }
        const int mt = x6_row_tiles(scans, n_scans, maxgroups);
        if (!mt) return FN_E_UNSUPPORTED;
        PArgs a;
        a.n = n_scans; a.H = H; a.no_hand = 0; a.spread = (scans[0].variant & 0x1000) ? 1 : 0;
        a.sync = reinterpret_cast<u32*>(scans[0].sync_ws);
        int groups = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruFwd& d = scans[s];
            PScan& f = a.s[s];
            f.w_frag = d.w_hh_frag; f.b_hh = d.b_hh; f.b_ih = d.b_ih; f.h0 = d.h0;
            f.gx_dense = d.gx_dense; f.gx_table = d.gx_table; f.idx = d.idx; f.gx_rowbias = d.gx_rowbias;
            f.h_all = d.h_all; f.gates = d.gates; f.xf = d.frag_ws; f.h0f = d.h0_frag; f.hlf = d.h_last_frag;
            if (d.h0 && !d.h0_frag) {                   // the initial state as triples into slab 0 (what step 0 reads)
                const int rc = fn_frag3_pack(d.h0, d.B, d.H, d.H, d.frag_ws, st);
                if (rc != FN_OK) return rc;
            }
            f.idx_ld = d.idx_ld; f.idx_shift = d.idx_shift; f.start_token = d.start_token; f.reverse = d.reverse;
            f.B = d.B; f.T = d.T;
            f.group0 = groups;
            groups += d.B / (64 * mt);
        }
        a.ngroups = groups;
        a.err = scans[0].err_ws ? reinterpret_cast<u32*>(scans[0].err_ws) : a.sync + FN_MAX_GROUPS * 32;
        if (!(scans[0].variant & 0x200)) {
            hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)FN_MAX_GROUPS * 32 * 4, st);
            if (me != hipSuccess) return (int)me;
        }
        const size_t lds = (size_t)3 * 16 * 3 * 1024 + (size_t)4 * 3 * RT * 4 + 16;
        // ping-pong form (kloop3_asm.h): exactly one input source per scan; variant bit 15 keeps the single-group kernel (tests)
        bool pp = !(scans[0].variant & 0x8000) && 2 * groups <= FN_MAX_GROUPS;
        for (int s = 0; s < n_scans && pp; ++s) pp = (scans[s].gx_table != nullptr) != (scans[s].gx_dense != nullptr);
        int rc;
        if (pp) rc = mt == 1 ? launch_k<PArgs, gru_fwd_x6pp_kernel<2>>(a, groups * nslices, lds, cus, st)
                             : launch_k<PArgs, gru_fwd_x6pp_kernel<1>>(a, groups * nslices, lds, cus, st);
        else rc = mt == 1 ? launch_k<PArgs, gru_fwd_x6_kernel<1>>(a, groups * nslices, lds, cus, st)
                          : launch_k<PArgs, gru_fwd_x6_kernel<2>>(a, groups * nslices, lds, cus, st);
        return rc == FN_PERSIST_NA ? FN_E_UNSUPPORTED : rc;
    }
    // smallest row block whose group count fits on the chip with one workgroup per CU
    int rpw = 0;
    const int cand[4] = {16, 32, 64, 128};
    const int force_rows = scans[0].variant & 0xFF;         // tuning / tests: take this row block or none
    for (int c = 0; c < 4 && !rpw; ++c) {
        if (force_rows && force_rows != cand[c]) continue;
        long groups = 0;
        for (int s = 0; s < n_scans; ++s) groups += (scans[s].B + cand[c] - 1) / cand[c];
        if (groups <= maxgroups) rpw = cand[c];
    }
    if (!rpw) return FN_PERSIST_NA;

    PArgs a;
    a.n = n_scans;
    a.H = H;
    a.no_hand = (scans[0].variant & 0x400) ? 1 : 0;
    a.spread = (scans[0].variant & 0x1000) ? 1 : 0;
    a.sync = reinterpret_cast<u32*>(scans[0].sync_ws);
    int groups = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruFwd& d = scans[s];
        PScan& f = a.s[s];
        f.w_frag = d.w_hh_frag; f.b_hh = d.b_hh; f.b_ih = d.b_ih; f.h0 = d.h0;
        f.gx_dense = d.gx_dense; f.gx_table = d.gx_table; f.idx = d.idx; f.gx_rowbias = d.gx_rowbias;
        f.h_all = d.h_all; f.gates = d.gates; f.xf = d.frag_ws; f.h0f = d.h0_frag; f.hlf = d.h_last_frag;
        f.idx_ld = d.idx_ld; f.idx_shift = d.idx_shift; f.start_token = d.start_token; f.reverse = d.reverse;
        f.B = d.B; f.T = d.T;
        f.group0 = groups;
        groups += (d.B + rpw - 1) / rpw;
        if (d.h0 && !d.h0_frag) {
            const int rc = launch_pack(d.h0, d.B, d.H, d.H, d.frag_ws, st);
            if (rc != FN_OK) return rc;
        }
    }
    a.ngroups = groups;
    a.err = scans[0].err_ws ? reinterpret_cast<u32*>(scans[0].err_ws) : a.sync + FN_MAX_GROUPS * 32;
    if (!(scans[0].variant & 0x200)) {                      // bit 9: the caller hands over counters that are already zero
        hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)FN_MAX_GROUPS * 32 * 4, st);      // counters only: err is sticky
        if (me != hipSuccess) return (int)me;
    }
    const int grid = groups * nslices;
    const int wk = rpw == 16 ? 4 : rpw == 32 ? 2 : (rpw == 64 && !(scans[0].variant & 0x100)) ? 2 : 1;     // K split of the chosen tiling
    const size_t lds = ((size_t)3 * H * 16 + (size_t)wk * (rpw / 16) * 3 * RT) * 4 + 16;
    // ping-pong form (two halves per workgroup, kloop2_asm.h): H = 512, 128- or 64-row groups, full row groups, saved gates, one input source
    bool pp = H == 512 && (rpw == 128 || (rpw == 64 && wk == 2)) && !(scans[0].variant & 0xC00) && 2 * groups <= FN_MAX_GROUPS;
    for (int s = 0; s < n_scans && pp; ++s) {
        const FnGruFwd& d = scans[s];
        pp = d.B % rpw == 0 && d.gates && d.T >= 2 && ((d.gx_table != nullptr) != (d.gx_dense != nullptr));
    }
    if (pp) return rpw == 128 ? launch_k<PArgs, gru_fwd_pp_kernel<1>>(a, grid, lds, cus, st) : launch_k<PArgs, gru_fwd_pp_kernel<2>>(a, grid, lds, cus, st);
    switch (rpw) {
        case 128: return launch_k<PArgs, gru_fwd_persist_kernel<4, 1, 2, 4>>(a, grid, lds, cus, st);
        case 64:
            if (scans[0].variant & 0x100) return launch_k<PArgs, gru_fwd_persist_kernel<4, 1, 1, 4>>(a, grid, lds, cus, st);
            return launch_k<PArgs, gru_fwd_persist_kernel<2, 2, 2, 4>>(a, grid, lds, cus, st);
        case 32: return launch_k<PArgs, gru_fwd_persist_kernel<2, 2, 1, 4>>(a, grid, lds, cus, st);
        default: return launch_k<PArgs, gru_fwd_persist_kernel<1, 4, 1, 4>>(a, grid, lds, cus, st);
    }
}

int fn_gru_bwd_persist(const FnGruBwd* scans, int n_scans, hipStream_t st) {
    const int H = scans[0].H;
    if (H > 512 || !scans[0].sync_ws) return FN_PERSIST_NA;
    int Tmax = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruBwd& d = scans[s];
        if (d.H != H) return FN_PERSIST_NA;
        const uintptr_t al = (uintptr_t)d.h0 | (uintptr_t)d.h_all | (uintptr_t)d.gates | (uintptr_t)d.dh_last | (uintptr_t)d.dh_ext |
                             (uintptr_t)d.dgx_all | (uintptr_t)d.dghn_all | (uintptr_t)d.dh0 | (uintptr_t)d.dgx_rowsum | (uintptr_t)d.dghn_rowsum;
        if (al & 15) return FN_PERSIST_NA;
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    if (Tmax < 2) return FN_PERSIST_NA;
    int cus = cu_count();
    if (scans[0].cu_budget > 0 && scans[0].cu_budget < cus) cus = scans[0].cu_budget;
    const int nslices = H / 16;
    if (cus <= 0 || nslices > cus) return FN_PERSIST_NA;
    const int force_rows = scans[0].variant & 0xFF;         // tuning / tests: take this row block or none
    if (scans[0].variant & 0x4000) {
        // exact split products on the bf16 MFMA (gru_bwd_x6_kernel): w_hh_t_frag = the bf16 triple image of W_hh^T (fn_weight_images kind 4), frag_ws of
        // 3 * fn_frag_floats(B, 3H) floats (the gate gradients are exchanged as triples).  Eligibility = fn_gru_bwd_x6_ok.
        const int th = x6_bwd_tiles(scans, n_scans, cus);
        if (!th) return FN_E_UNSUPPORTED;
        QArgs a;
        a.n = n_scans; a.H = H; a.no_hand = 0; a.spread = (scans[0].variant & 0x1000) ? 1 : 0;
        a.sync = reinterpret_cast<u32*>(scans[0].sync_ws);
        int groups = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruBwd& d = scans[s];
            QScan& f = a.s[s];
            f.wt_frag = d.w_hh_t_frag; f.h0 = d.h0; f.h_all = d.h_all; f.gates = d.gates;
            f.dh_last = d.dh_last; f.dh_ext = d.dh_ext;
            f.dgx_all = d.dgx_all; f.dghn_all = d.dghn_all; f.dh0 = d.dh0;
            f.rowsum = d.dgx_rowsum; f.rowsum_n = d.dghn_rowsum; f.xf = d.frag_ws;
            f.B = d.B; f.T = d.T;
            f.group0 = groups;
            groups += d.B / (32 * th);
        }
        a.ngroups = groups;
        a.err = scans[0].err_ws ? reinterpret_cast<u32*>(scans[0].err_ws) : a.sync + FN_MAX_GROUPS * 32;
        if (!(scans[0].variant & 0x200)) {
            hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)FN_MAX_GROUPS * 32 * 4, st);
            if (me != hipSuccess) return (int)me;
        }
        const size_t lds = (size_t)4 * 11 * 3072 + (size_t)4 * (2 * th) * RT * 4 + 16;
        const int rc = th == 2 ? launch_k<QArgs, gru_bwd_x6_kernel<2>>(a, groups * 16, lds, cus, st) : launch_k<QArgs, gru_bwd_x6_kernel<1>>(a, groups * 16, lds, cus, st);
        return rc == FN_PERSIST_NA ? FN_E_UNSUPPORTED : rc;
    }
    // Register-stationary ping-pong form (gru_bwd_rs_kernel): H = 512, 16 slices of 32 columns x up to 16 groups of 64 (or 32) rows, every group
    // full, more than half of the chip used (smaller problems keep the 32-slice kernels).  Variant bits 10 / 11 / 13 select the older loops.
    if (H == 512 && !(scans[0].variant & 0x2C00) && !force_rows) {
        long g64 = 0, g32 = 0;
        bool d64 = true, d32 = true;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruBwd& d = scans[s];
            g64 += d.B / 64; g32 += d.B / 32;
            d64 = d64 && d.B % 64 == 0 && d.T >= 2;
            d32 = d32 && d.B % 32 == 0 && d.T >= 2;
        }
        const int th = (d64 && g64 > 8 && g64 <= 16 && g64 * 16 <= cus) ? 2 : (d32 && g32 > 8 && g32 <= 16 && g32 * 16 <= cus) ? 1 : 0;
        if (th) {
            QArgs a;
            a.n = n_scans; a.H = H; a.no_hand = 0; a.spread = (scans[0].variant & 0x1000) ? 1 : 0;
            a.sync = reinterpret_cast<u32*>(scans[0].sync_ws);
            int groups = 0;
            for (int s = 0; s < n_scans; ++s) {
                const FnGruBwd& d = scans[s];
                QScan& f = a.s[s];
                f.wt_frag = d.w_hh_t_frag; f.h0 = d.h0; f.h_all = d.h_all; f.gates = d.gates;
                f.dh_last = d.dh_last; f.dh_ext = d.dh_ext;
                f.dgx_all = d.dgx_all; f.dghn_all = d.dghn_all; f.dh0 = d.dh0;
                f.rowsum = d.dgx_rowsum; f.rowsum_n = d.dghn_rowsum; f.xf = d.frag_ws;
                f.B = d.B; f.T = d.T;
                f.group0 = groups;
                groups += d.B / (32 * th);
            }
            a.ngroups = groups;
            a.err = scans[0].err_ws ? reinterpret_cast<u32*>(scans[0].err_ws) : a.sync + FN_MAX_GROUPS * 32;
            if (!(scans[0].variant & 0x200)) {
                hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)FN_MAX_GROUPS * 32 * 4, st);
                if (me != hipSuccess) return (int)me;
            }
            const size_t lds = ((size_t)3 * H * 16 + (size_t)2 * 4 * (2 * th) * RT) * 4 + 16;
            const int rc = th == 2 ? launch_k<QArgs, gru_bwd_rs_kernel<2>>(a, groups * 16, lds, cus, st) : launch_k<QArgs, gru_bwd_rs_kernel<1>>(a, groups * 16, lds, cus, st);
            if (rc != FN_PERSIST_NA) return rc;
        }
    }
    const int maxgroups = cus / nslices < FN_MAX_GROUPS ? cus / nslices : FN_MAX_GROUPS;
    int rpw = 0;
    const int cand[4] = {16, 32, 64, 128};
    for (int c = 0; c < 4 && !rpw; ++c) {
        if (force_rows && force_rows != cand[c]) continue;
        long groups = 0;
        for (int s = 0; s < n_scans; ++s) groups += (scans[s].B + cand[c] - 1) / cand[c];
        if (groups <= maxgroups) rpw = cand[c];
    }
    if (!rpw) return FN_PERSIST_NA;

    QArgs a;
    a.n = n_scans;
    a.H = H;
    a.no_hand = (scans[0].variant & 0x400) ? 1 : 0;
    a.spread = (scans[0].variant & 0x1000) ? 1 : 0;
    a.sync = reinterpret_cast<u32*>(scans[0].sync_ws);
    int groups = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruBwd& d = scans[s];
        QScan& f = a.s[s];
        f.wt_frag = d.w_hh_t_frag; f.h0 = d.h0; f.h_all = d.h_all; f.gates = d.gates;
        f.dh_last = d.dh_last; f.dh_ext = d.dh_ext;
        f.dgx_all = d.dgx_all; f.dghn_all = d.dghn_all; f.dh0 = d.dh0;
        f.rowsum = d.dgx_rowsum; f.rowsum_n = d.dghn_rowsum; f.xf = d.frag_ws;
        f.B = d.B; f.T = d.T;
        f.group0 = groups;
        groups += (d.B + rpw - 1) / rpw;
    }
    a.ngroups = groups;
    a.err = scans[0].err_ws ? reinterpret_cast<u32*>(scans[0].err_ws) : a.sync + FN_MAX_GROUPS * 32;
    if (!(scans[0].variant & 0x200)) {
        hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)FN_MAX_GROUPS * 32 * 4, st);
        if (me != hipSuccess) return (int)me;
    }
    const int grid = groups * nslices;
    const int wk = rpw == 16 ? 4 : rpw == 32 ? 2 : (rpw == 64 && !(scans[0].variant & 0x100)) ? 2 : 1;
    const size_t lds = ((size_t)3 * H * 16 + (size_t)2 * wk * (rpw / 16) * RT) * 4 + 16;
    switch (rpw) {
        case 128: return launch_k<QArgs, gru_bwd_persist_kernel<4, 1, 2, 8>>(a, grid, lds, cus, st);
        case 64:
            if (scans[0].variant & 0x100) return launch_k<QArgs, gru_bwd_persist_kernel<4, 1, 1, 8>>(a, grid, lds, cus, st);
            return launch_k<QArgs, gru_bwd_persist_kernel<2, 2, 2, 8>>(a, grid, lds, cus, st);
        case 32: return launch_k<QArgs, gru_bwd_persist_kernel<2, 2, 1, 8>>(a, grid, lds, cus, st);
        default: return launch_k<QArgs, gru_bwd_persist_kernel<1, 4, 1, 8>>(a, grid, lds, cus, st);
    }
}
