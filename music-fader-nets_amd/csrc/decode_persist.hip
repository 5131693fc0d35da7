// decode_persist.hip - greedy autoregressive decode of the global decoder (gmm_model.py:119-149 with model.eval()) for up to 2048
// sequences as ONE launch: the per-token chain  layer-1 cell -> W_ih2 projection -> layer-2 cell -> 512->V output layer ->
// log-softmax + argmax -> next token  is latency-bound (five dependent kernels per token otherwise), so each link gets its own set
// of workgroups that keep their weight slice in LDS for the whole decode and hand the activations to the next set through L2:
//
//   L1  (H/16 workgroups)  W_hh1 slice [48 x H]    waits C1(t-1), then C4(t-1): own argmax -> h1_t slice -> X1[t&1]  arrive C1
//   P2  (H/16)             W_ih2 slice [48 x H]    waits C1                   -> gx2 slice    -> G2        arrive C2
//   L2  (H/16)             W_hh2 slice [48 x H]    waits C3(t-1) [C1 at t=0], then C2 -> h2_t -> X2[t&1]   arrive C3
//   OUT (ceil(V/16))       W_out slice [16 x H]    waits C3                   -> logits slice -> LOGITS    arrive C4
//   ARG (1)                -                       waits C4                   -> log-softmax, argmax -> tokens / logp out, arrive C5
//                                                   (OUT waits C5(t-1) before it overwrites the logits)
//
// Every hand-over is recipe R1 of the guide (write-through sc1 payload, every wave drains vmcnt, barrier, one relaxed agent-scope
// atomic on a monotonic counter; consumer: one lane polls, barrier, sc1 loads); every spin is bounded (timeout -> sticky error word,
// all workgroups leave).  The ping-pong slabs cannot be overwritten early: a writer of step t+2 only starts after token t+1 exists,
// which is after every reader of step t has arrived.  The 4 waves of a workgroup split K and add their partial tiles through LDS.
//
// More than 32 sequences (the reference's evaluator decodes 8 fader values x 100 samples = 800 rows at once, test_class.py:84-85): the
// batch is cut into BLOCKS of 32 rows (64 rows from FN_DECODE_MT4_ROWS sequences on) that travel through the same role workgroups one after the other - a pipeline: while L1 works on
// block b+1, P2 has block b, L2 block b-1 ... - with one set of counters and exchange slabs per block; the LDS-resident weight slices are
// reused by every block, and a second replica of the whole role set (2 x 119 workgroups <= 256 CUs at H = 512) takes every other block.
// Per-block state (a cell's own previous state slice, the per-row input constants) is re-read from the exchange slabs / L2 instead of
// living in registers.  Per token: max(chain latency of one block, blocks per replica x busy time of the slowest role).
#include <atomic>

#include "gru_layout.h"

#ifdef FN_TIMING
// cycle stamps of the role workgroups at step 10 of the first block (slice 0 of each role): scratch/timing_decode.py
__device__ unsigned long long fn_ddbg[5 * 16];
#define FN_DSTAMP(role_, k)                                                                                              \
    do {                                                                                                                 \
        if (slice == 0 && rep == 0 && blk == 0 && threadIdx.x == 0 && t == 10) fn_ddbg[(role_) * 16 + (k)] = __builtin_readcyclecounter(); \
    } while (0)
// the same role's NEXT block of the same step (period of the role's block loop = its busy + idle time per block)
#define FN_DSTAMP_NEXT(role_)                                                                                            \
    do {                                                                                                                 \
        if (slice == 0 && rep == 0 && blk == a.nrep && threadIdx.x == 0 && t == 10) fn_ddbg[(role_) * 16 + 8] = __builtin_readcyclecounter(); \
    } while (0)
extern "C" int fn_ddbg_read(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(fn_ddbg), sizeof(unsigned long long) * 80);
}
#else
#define FN_DSTAMP(role_, k)
#define FN_DSTAMP_NEXT(role_)
#endif

namespace {

constexpr int NT = 256;
constexpr int C1 = 0, C2 = 32, C3 = 64, C4 = 96, C5 = 128;               // word offsets inside a block's counters (one 128-byte line each)
constexpr int BLKW = 160;                                                 // counter words per block
#ifndef FN_DECODE_MT4_ROWS
#define FN_DECODE_MT4_ROWS 353                                            // from this many sequences on: 64-row blocks (measured crossover, profiles/r03_decode_block_pipeline.txt: 320 rows 34.5 vs 36.4 us per token, 384 rows 41.8 vs 35.9)
#endif
constexpr int MAXBLK = 32;                                                // blocks of 32 / 64 rows: up to 1024 / 2048 sequences
constexpr int ERRW = MAXBLK * BLKW;                                       // sticky error word behind all counters

struct DArgs {
    int B, steps, H, V, nvt;                         // nvt = ceil(V / 16) output slices
    int nblk, nrep;                                  // row blocks (of MT x 16 rows) and replicas of the role set (replica r: blocks r, r + nrep, ...)
    int start_token, tok_ld;
    const float *w1, *bhh1, *bih1, *table1, *rowbias1, *h0;
    const float *wi2, *bih2, *w2, *bhh2;
    const float *wo, *bo;
    float *x1, *x2, *g2, *logits;
    int* tokens;
    float* logp;                                     // [B][steps][V] or null
    u32* sync;
};

// L1-bypassing dword load WITHOUT a wait (pair with fn_wait_vm<0>()): an agent-scope atomic load is followed by its own s_waitcnt,
// 48 of them per block were 48 dependent L2 round trips (13.8 k of the layer-1 role's 28 k cycles per block, scratch/timing_decode.py)
FN_DEVINL void gld1_sc1(float& dst, const float* p) { asm volatile("global_load_dword %0, %1, off sc1" : "=v"(dst) : "v"(p) : "memory"); }

FN_DEVINL f32x4 ldv4_sc1(const float* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

struct Sync {
    u32* base;                                       // this block's counters
    u32* errw;                                       // the launch's sticky error word
    volatile int* dead;
    // one lane waits until *counter >= target, then the whole workgroup continues; false = give up (bounded spin / error elsewhere)
    FN_DEVINL bool wait(int counter, u32 target) const {
        if (threadIdx.x == 0) {
            u32 spins = 0;
            while (ld_cnt(base + counter) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0 && (spins > SPIN_LIMIT || ld_cnt(errw) != 0)) {
                    __hip_atomic_store(errw, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    *dead = 1;
                    break;
                }
            }
        }
        __syncthreads();
        return *dead == 0;
    }
    // publish: every wave has drained its stores, then one lane arrives
    FN_DEVINL void arrive(int counter) const {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(base + counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

// acc[m][n] (this wave's K quarter) = X[rows of tile m][k] * Wslice[n][k]; X fragment-major in global memory, W in LDS
template <int MT, int NTN>
FN_DEVINL void kquarter(const float* xin, const float* wl, int nk, int lane, int wave, f32x4 (&acc)[MT][NTN]) {
    const int c0 = nk * wave / 4, nkw = nk * (wave + 1) / 4 - c0;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTN; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // at most 4 chunks per wave (H <= 512): all operand loads go out first, ONE wait, then the MFMAs
    f32x4 fa[4][MT][2];
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it < nkw) {
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                gld4_sc1(fa[it][m][0], xin + ((long)m * nk + c0 + it) * 512 + lane * 4);
                gld4_sc1(fa[it][m][1], xin + ((long)m * nk + c0 + it) * 512 + 256 + lane * 4);
            }
        }
    // four chunks requested (H = 512): chunk `it` is multiplied as soon as it has landed, the younger ones stay in flight behind the MFMAs
    const bool full = nkw == 4;
    if (!full) fn_wait_vm<0>();
#pragma unroll
    for (int it = 0; it < 4; ++it)
        if (it < nkw) {
            if (full) {
                if (it == 0) fn_wait_vm<3 * MT * 2>();
                else if (it == 1) fn_wait_vm<2 * MT * 2>();
                else if (it == 2) fn_wait_vm<1 * MT * 2>();
                else fn_wait_vm<0>();
            }
            const int c = c0 + it;
            f32x4 fb[NTN][2];
#pragma unroll
            for (int n = 0; n < NTN; ++n) {
                fb[n][0] = *reinterpret_cast<const f32x4*>(wl + ((long)(n * nk + c) * 2 + 0) * 256 + lane * 4);
                fb[n][1] = *reinterpret_cast<const f32x4*>(wl + ((long)(n * nk + c) * 2 + 1) * 256 + lane * 4);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < NTN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(f4at(fa[it][m][j >> 2], j & 3), f4at(fb[n][j >> 2], j & 3), acc[m][n], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MT; ++m) { fn_keep(fa[it][m][0]); fn_keep(fa[it][m][1]); }
        }
}

template <int MT, int NTN>
FN_DEVINL void spill_partials(float* red, int lane, int wave, const f32x4 (&acc)[MT][NTN]) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NTN; ++n)
            *reinterpret_cast<f32x4*>(red + (long)((wave * MT + m) * NTN + n) * RT + lane * 4 + (lane >> 4) * 4) = acc[m][n];
    __syncthreads();
}

// sum over the 4 K quarters of accumulator (tile, n) at item (row rl, units 4 u4 ..)
template <int MT, int NTN>
FN_DEVINL f32x4 gather_sum(const float* red, int tile, int n, int coff) {
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int c = 0; c < 4; ++c) s[c] += red[(long)((w * MT + tile) * NTN + n) * RT + coff + c * 4];
    return s;
}

template <int MT>
__global__ __launch_bounds__(NT) void decode_greedy_kernel(const DArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = a.H, nk = H >> 5, nsl = H >> 4, B = a.B;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* wl = smem;                                 // weight slice, B-fragment order
    float* red = smem + 3 * H * 16;                   // [4][MT][3][RT]
    volatile int* dead = reinterpret_cast<volatile int*>(red + 4 * MT * 3 * RT);
    volatile int* tokl = dead + 4;                    // [64] tokens of the current step (layer-1 role)
    if (tid == 0) *dead = 0;
    const long FS = (long)MT * 16 * H;                // floats of one block's exchange slab
    const long FSA = FS * a.nblk;                     // ... of one slot (all blocks)
    constexpr int RB = MT * 16;                       // rows per block

    int role, slice;
    const int per_rep = 3 * nsl + a.nvt + 1;
    const int rep = blockIdx.x / per_rep;
    {
        const int b = blockIdx.x % per_rep;
        if (b < nsl) { role = 0; slice = b; }
        else if (b < 2 * nsl) { role = 1; slice = b - nsl; }
        else if (b < 3 * nsl) { role = 2; slice = b - 2 * nsl; }
        else if (b < 3 * nsl + a.nvt) { role = 3; slice = b - 3 * nsl; }
        else { role = 4; slice = 0; }
    }
    const bool single = a.nblk <= a.nrep;             // one block per replica: a cell keeps its state slice in registers
    // one-time: weight slice -> LDS
    if (role < 3) {
        const float* w = role == 0 ? a.w1 : (role == 1 ? a.wi2 : a.w2);
#pragma unroll 1
        for (int q = 0; q < 3; ++q) {
            const float4* src = reinterpret_cast<const float4*>(w + (long)(q * nsl + slice) * nk * 512);
            float4* dst = reinterpret_cast<float4*>(wl + (long)q * nk * 512);
            for (int i = tid; i < nk * 128; i += NT) dst[i] = src[i];
        }
    } else if (role == 3) {
        const float4* src = reinterpret_cast<const float4*>(a.wo + (long)slice * nk * 512);
        float4* dst = reinterpret_cast<float4*>(wl);
        for (int i = tid; i < nk * 128; i += NT) dst[i] = src[i];
    }
    __syncthreads();

    // epilogue item of this thread: row rl of the BLOCK, units / columns 4 u4 .. of the slice
    const int item = tid, rl = item >> 2, u4 = item & 3;
    const int tile = min(rl >> 4, MT - 1);
    const int coff = ((rl & 15) >> 2) * 68 + u4 * 16 + (rl & 3);
    const int jj0 = slice * 16 + 4 * u4;              // hidden unit (roles 0-2) or vocabulary column (role 3)
    const u32 unsl = (u32)nsl;
    const int vpad = a.nvt * 16;
    u32* errw = a.sync + ERRW;

    if (role == 0) {                                  // ---------------- layer-1 cell ----------------
        f32x4 bh[3], bi[3], hp_reg = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bh[q] = ldv4(a.bhh1 + q * H + jj0);
            bi[q] = a.bih1 ? ldv4(a.bih1 + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        for (int t = 0; t < a.steps; ++t) {
            for (int blk = rep; blk < a.nblk; blk += a.nrep) {
                const Sync sy = {a.sync + blk * BLKW, errw, dead};
                const int r0 = blk * RB, nrow = min(RB, B - r0);          // rows of this block
                const bool act = item < MT * 64 && rl < nrow;
                const int b = r0 + min(rl, nrow - 1);
                float* x1 = a.x1 + (long)blk * FS;
                FN_DSTAMP(0, 0);
                FN_DSTAMP_NEXT(0);
                // the recurrent product needs h1_{t-1} only, not the token: it runs while the previous token is still being produced
                if (t > 0 && !sy.wait(C1, unsl * (u32)t)) return;
                FN_DSTAMP(0, 1);
                const float* xin = x1 + (long)((t + 1) & 1) * FSA;        // slot 1 holds h0 at t = 0
                // own previous state slice and the per-row input constants: registers when this workgroup serves ONE block, else re-read
                // (requested together with the operand fragments: kquarter's single wait covers the asm load)
                f32x4 hp = hp_reg, rb[3];
                const bool reload = !(single && t > 0);
                if (reload) gld4_sc1(hp, t == 0 ? a.h0 + (long)b * H + jj0 : xin + frag_off(min(rl, nrow - 1), jj0, nk));
#pragma unroll
                for (int q = 0; q < 3; ++q) rb[q] = a.rowbias1 ? ldv4(a.rowbias1 + (long)b * 3 * H + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 acc[MT][3];
                kquarter<MT, 3>(xin, wl, nk, lane, wave, acc);
                fn_touch(hp);
                FN_DSTAMP(0, 2);
                spill_partials<MT, 3>(red, lane, wave, acc);
                f32x4 gh[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) gh[q] = gather_sum<MT, 3>(red, tile, q, coff) + bh[q];
                FN_DSTAMP(0, 3);
                // the token of step t-1: every layer-1 workgroup takes the argmax of the logits itself as soon as the output slices
                // have arrived (the ARG workgroup, which writes tokens and log-probabilities out, is off the critical chain)
                int tok = a.start_token;
                if (t > 0 && !single) {
                    // throughput regime (several blocks per replica): the token comes from the ARG workgroup - one more hand-over in a block's
                    // chain, but the 32-fold redundant argmax (12 k of this role's 26 k cycles per block) no longer bounds the pipeline
                    if (!sy.wait(C5, (u32)t)) return;
                    tok = __hip_atomic_load(a.tokens + (long)b * a.tok_ld + (t - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    FN_DSTAMP(0, 5);
                } else if (t > 0) {
                    if (!sy.wait(C4, (u32)a.nvt * (u32)t)) return;
                    FN_DSTAMP(0, 4);
                    // all rows of this wave are requested before the first is reduced: one L2 round trip instead of one per row
                    constexpr int RPW = MT * 4;                              // rows per wave
                    float xl[RPW][6];
#pragma unroll
                    for (int j = 0; j < RPW; ++j) {
                        const int rowc = min(wave + 4 * j, nrow - 1);
#pragma unroll
                        for (int k = 0; k < 6; ++k) gld1_sc1(xl[j][k], a.logits + (long)(r0 + rowc) * vpad + min(lane + 64 * k, vpad - 1));
                    }
                    fn_wait_vm<0>();
#pragma unroll
                    for (int j = 0; j < RPW; ++j)
#pragma unroll
                        for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(xl[j][k]));
                    // the cross-lane rounds of ALL rows side by side (round-major): 8 independent ds_bpermute chains in flight instead of 8
                    // dependent chains one after the other (that order was 14 k of the role's 28 k cycles per block, scratch/timing_decode.py)
                    float mxr[RPW];
                    int amr[RPW];
#pragma unroll
                    for (int j = 0; j < RPW; ++j) {
                        mxr[j] = -3.0e38f;
                        amr[j] = 0x7fffffff;
#pragma unroll
                        for (int k = 0; k < 6; ++k) {
                            const int v = lane + 64 * k;
                            const float x = v < a.V ? xl[j][k] : -3.0e38f;
                            if (x > mxr[j]) { mxr[j] = x; amr[j] = v; }
                        }
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        float om[RPW];
                        int oa[RPW];
#pragma unroll
                        for (int j = 0; j < RPW; ++j) { om[j] = __shfl_xor(mxr[j], o, 64); oa[j] = __shfl_xor(amr[j], o, 64); }
#pragma unroll
                        for (int j = 0; j < RPW; ++j)
                            if (om[j] > mxr[j] || (om[j] == mxr[j] && oa[j] < amr[j])) { mxr[j] = om[j]; amr[j] = oa[j]; }
                    }
#pragma unroll
                    for (int j = 0; j < RPW; ++j)
                        if (lane == 0 && wave + 4 * j < nrow) tokl[wave + 4 * j] = amr[j];
                    __syncthreads();
                    tok = tokl[min(rl, nrow - 1)];
                    FN_DSTAMP(0, 5);
                }
                f32x4 ex[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) ex[q] = ldv4(a.table1 + (long)tok * 3 * H + q * H + jj0);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float r = fn_sigmoid(((bi[0][c] + ex[0][c]) + rb[0][c]) + gh[0][c]);
                    const float z = fn_sigmoid(((bi[1][c] + ex[1][c]) + rb[1][c]) + gh[1][c]);
                    const float n = fn_tanh(((bi[2][c] + ex[2][c]) + rb[2][c]) + r * gh[2][c]);
                    hp[c] = (1.0f - z) * n + z * hp[c];
                }
                hp_reg = hp;
                if (act) stv4_sc1(x1 + (long)(t & 1) * FSA + frag_off(rl, jj0, nk), hp);
                FN_DSTAMP(0, 6);
                sy.arrive(C1);
                FN_DSTAMP(0, 7);
            }
        }
    } else if (role == 1) {                           // ---------------- W_ih2 projection ----------------
        f32x4 bi[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) bi[q] = a.bih2 ? ldv4(a.bih2 + q * H + jj0) : (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < a.steps; ++t) {
            for (int blk = rep; blk < a.nblk; blk += a.nrep) {
                const Sync sy = {a.sync + blk * BLKW, errw, dead};
                const int r0 = blk * RB, nrow = min(RB, B - r0);
                const bool act = item < MT * 64 && rl < nrow;
                const int b = r0 + min(rl, nrow - 1);
                // (g2 of step t-1 is free: h1_t exists only after token t-1, i.e. after layer 2 consumed it)
                FN_DSTAMP(1, 0);
                FN_DSTAMP_NEXT(1);
                if (!sy.wait(C1, unsl * (u32)(t + 1))) return;
                FN_DSTAMP(1, 1);
                f32x4 acc[MT][3];
                kquarter<MT, 3>(a.x1 + (long)blk * FS + (long)(t & 1) * FSA, wl, nk, lane, wave, acc);
                spill_partials<MT, 3>(red, lane, wave, acc);
                if (act) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) stv4_sc1(a.g2 + (long)b * 3 * H + q * H + jj0, gather_sum<MT, 3>(red, tile, q, coff) + bi[q]);
                }
                sy.arrive(C2);
                FN_DSTAMP(1, 7);
            }
        }
    } else if (role == 2) {                           // ---------------- layer-2 cell ----------------
        f32x4 bh[3], hp_reg = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < 3; ++q) bh[q] = ldv4(a.bhh2 + q * H + jj0);
        for (int t = 0; t < a.steps; ++t) {
            for (int blk = rep; blk < a.nblk; blk += a.nrep) {
                const Sync sy = {a.sync + blk * BLKW, errw, dead};
                const int r0 = blk * RB, nrow = min(RB, B - r0);
                const bool act = item < MT * 64 && rl < nrow;
                const int b = r0 + min(rl, nrow - 1);
                // recurrent input: h2_{t-1}, or h1_0 at the first step (gmm_model.py:134-135: hx[1] = hx[0] at i == 0)
                FN_DSTAMP(2, 0);
                FN_DSTAMP_NEXT(2);
                if (!(t == 0 ? sy.wait(C1, unsl) : sy.wait(C3, unsl * (u32)t))) return;
                FN_DSTAMP(2, 1);
                const float* xin = t == 0 ? a.x1 + (long)blk * FS : a.x2 + (long)blk * FS + (long)((t + 1) & 1) * FSA;
                f32x4 hp = hp_reg;
                if (!(single && t > 0)) gld4_sc1(hp, xin + frag_off(min(rl, nrow - 1), jj0, nk));
                f32x4 acc[MT][3];
                kquarter<MT, 3>(xin, wl, nk, lane, wave, acc);
                fn_touch(hp);
                spill_partials<MT, 3>(red, lane, wave, acc);
                f32x4 gh[3];
#pragma unroll
                for (int q = 0; q < 3; ++q) gh[q] = gather_sum<MT, 3>(red, tile, q, coff) + bh[q];
                FN_DSTAMP(2, 3);
                if (!sy.wait(C2, unsl * (u32)(t + 1))) return;
                FN_DSTAMP(2, 4);
                f32x4 gx[3];                                              // three requests, ONE wait
#pragma unroll
                for (int q = 0; q < 3; ++q) gld4_sc1(gx[q], a.g2 + (long)b * 3 * H + q * H + jj0);
                fn_wait_vm<0>();
#pragma unroll
                for (int q = 0; q < 3; ++q) fn_touch(gx[q]);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float r = fn_sigmoid(gx[0][c] + gh[0][c]);
                    const float z = fn_sigmoid(gx[1][c] + gh[1][c]);
                    const float n = fn_tanh(gx[2][c] + r * gh[2][c]);
                    hp[c] = (1.0f - z) * n + z * hp[c];
                }
                hp_reg = hp;
                if (act) stv4_sc1(a.x2 + (long)blk * FS + (long)(t & 1) * FSA + frag_off(rl, jj0, nk), hp);
                sy.arrive(C3);
                FN_DSTAMP(2, 7);
            }
        }
    } else if (role == 3) {                           // ---------------- output layer slice ----------------
        f32x4 bo = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (jj0 + c < a.V) bo[c] = a.bo[jj0 + c];
        for (int t = 0; t < a.steps; ++t) {
            for (int blk = rep; blk < a.nblk; blk += a.nrep) {
                const Sync sy = {a.sync + blk * BLKW, errw, dead};
                const int r0 = blk * RB, nrow = min(RB, B - r0);
                const bool act = item < MT * 64 && rl < nrow;
                const int b = r0 + min(rl, nrow - 1);
                // the logits of step t-1 may only be overwritten once the ARG workgroup has read them (this wait is off the critical
                // chain: the output slices are idle until layer 2 arrives anyway); every layer-1 workgroup has taken its own argmax of them
                // before h1_t, hence h2_t, existed
                FN_DSTAMP(3, 0);
                FN_DSTAMP_NEXT(3);
                if (t > 0 && !sy.wait(C5, (u32)t)) return;
                if (!sy.wait(C3, unsl * (u32)(t + 1))) return;
                FN_DSTAMP(3, 1);
                f32x4 acc[MT][1];
                kquarter<MT, 1>(a.x2 + (long)blk * FS + (long)(t & 1) * FSA, wl, nk, lane, wave, acc);
                spill_partials<MT, 1>(red, lane, wave, acc);
                if (act) stv4_sc1(a.logits + (long)b * vpad + jj0, gather_sum<MT, 1>(red, tile, 0, coff) + bo);
                sy.arrive(C4);
                FN_DSTAMP(3, 7);
            }
        }
    } else {                                          // ---------------- log-softmax + first-index argmax ----------------
        for (int t = 0; t < a.steps; ++t) {
            for (int blk = rep; blk < a.nblk; blk += a.nrep) {
                const Sync sy = {a.sync + blk * BLKW, errw, dead};
                const int r0 = blk * RB, nrow = min(RB, B - r0);
                FN_DSTAMP(4, 0);
                FN_DSTAMP_NEXT(4);
                if (!sy.wait(C4, (u32)a.nvt * (u32)(t + 1))) return;
                FN_DSTAMP(4, 1);
                constexpr int RPW = MT * 4;
                float xl[RPW][6];
#pragma unroll
                for (int j = 0; j < RPW; ++j) {                               // every row of this wave requested up front
                    const int rowc = r0 + min(wave + 4 * j, nrow - 1);
#pragma unroll
                    for (int k = 0; k < 6; ++k) gld1_sc1(xl[j][k], a.logits + (long)rowc * vpad + min(lane + 64 * k, vpad - 1));
                }
                fn_wait_vm<0>();
#pragma unroll
                for (int j = 0; j < RPW; ++j)
#pragma unroll
                    for (int k = 0; k < 6; ++k) asm volatile("" : "+v"(xl[j][k]));
                float mxr[RPW];
                int amr[RPW];
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    mxr[j] = -3.0e38f;
                    amr[j] = 0x7fffffff;
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int v = lane + 64 * k;
                        if (v >= a.V) xl[j][k] = -3.0e38f;
                        if (xl[j][k] > mxr[j]) { mxr[j] = xl[j][k]; amr[j] = v; }     // ascending v: the first index wins inside a lane
                    }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {                            // round-major: the rows' cross-lane chains side by side
                    float om[RPW];
                    int oa[RPW];
#pragma unroll
                    for (int j = 0; j < RPW; ++j) { om[j] = __shfl_xor(mxr[j], o, 64); oa[j] = __shfl_xor(amr[j], o, 64); }
#pragma unroll
                    for (int j = 0; j < RPW; ++j)
                        if (om[j] > mxr[j] || (om[j] == mxr[j] && oa[j] < amr[j])) { mxr[j] = om[j]; amr[j] = oa[j]; }
                }
                float se[RPW];
                if (a.logp) {                                                // sum of exponentials, also round-major
#pragma unroll
                    for (int j = 0; j < RPW; ++j) {
                        se[j] = 0.f;
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            if (lane + 64 * k < a.V) se[j] += expf(xl[j][k] - mxr[j]);
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        float t2[RPW];
#pragma unroll
                        for (int j = 0; j < RPW; ++j) t2[j] = __shfl_xor(se[j], o, 64);
#pragma unroll
                        for (int j = 0; j < RPW; ++j) se[j] += t2[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < RPW; ++j) {
                    if (wave + 4 * j >= nrow) continue;
                    const int row = r0 + wave + 4 * j;
                    if (a.logp) {
                        const float lse = mxr[j] + logf(se[j]);
                        float* out = a.logp + ((long)row * a.steps + t) * a.V;
#pragma unroll
                        for (int k = 0; k < 6; ++k)
                            if (lane + 64 * k < a.V) out[lane + 64 * k] = xl[j][k] - lse;
                    }
                    if (lane == 0) __hip_atomic_store(a.tokens + (long)row * a.tok_ld + t, amr[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sy.arrive(C5);
                FN_DSTAMP(4, 7);
            }
        }
    }
}

template <int MT>
int launch_decode(const DArgs& a, int grid, hipStream_t st) {
    auto k = decode_greedy_kernel<MT>;
    static std::atomic<int> fits[32];              // write-once per device (zero-initialised): 1 = at least one workgroup of this kernel fits a CU, -1 = it does not
    const size_t lds = ((size_t)3 * a.H * 16 + (size_t)4 * MT * 3 * RT) * 4 + 16 + 64 * 4;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_UNSUPPORTED;
    if (fits[dev].load(std::memory_order_acquire) == 0) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 64);
        if (e != hipSuccess) return (int)e;
        int nb = 0;                                // every role workgroup must be resident (one per CU): ask the occupancy calculator
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(k), NT, 160 * 1024 - 64);
        if (e != hipSuccess) return (int)e;
        fits[dev].store(nb > 0 ? 1 : -1, std::memory_order_release);   // idempotent: racing threads compute the same value
    }
    if (fits[dev] < 0) return FN_E_UNSUPPORTED;
    hipLaunchKernelGGL(k, dim3(grid), dim3(NT), lds, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // namespace

extern "C" {

size_t fn_decode_ws_bytes(int B, int H, int V) {
    const size_t bp = B <= 16 ? 16 : (size_t)(B + 63) / 64 * 64, vp = (size_t)(V + 15) / 16 * 16;    // rows as the kernel tiles them (blocks of 16 / 32 / 64)
    return (4 * bp * H + bp * 3 * H + bp * vp) * sizeof(float);
}

size_t fn_decode_sync_ws_bytes(void) { return (size_t)(ERRW + 32) * 4; }

int fn_decode_greedy(const FnDecode* d, void* stream) {
    if (!d) return FN_E_NULL;
    if (!d->w_hh1_frag || !d->b_hh1 || !d->table1 || !d->h0 || !d->w_ih2_frag || !d->w_hh2_frag || !d->b_hh2 || !d->w_out_frag || !d->b_out ||
        !d->tokens || !d->ws || !d->sync_ws)
        return FN_E_NULL;
    if (d->B <= 0 || d->B > MAXBLK * 64 || d->steps <= 0 || d->H <= 0 || (d->H % 32) != 0 || d->H > 512 || d->V <= 0 || d->V > 384 || d->tok_ld < d->steps)
        return FN_E_SHAPE;
    const uintptr_t al = (uintptr_t)d->w_hh1_frag | (uintptr_t)d->w_ih2_frag | (uintptr_t)d->w_hh2_frag | (uintptr_t)d->w_out_frag |
                         (uintptr_t)d->b_hh1 | (uintptr_t)d->b_ih1 | (uintptr_t)d->table1 | (uintptr_t)d->rowbias1 | (uintptr_t)d->h0 |
                         (uintptr_t)d->b_ih2 | (uintptr_t)d->b_hh2 | (uintptr_t)d->ws;
    if (al & 15) return FN_E_ALIGN;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return FN_E_SHAPE;
    const int nsl = d->H / 16, nvt = (d->V + 15) / 16;
    const int per_rep = 3 * nsl + nvt + 1;
    if (per_rep > prop.multiProcessorCount) return FN_E_UNSUPPORTED;  // every workgroup must be resident: one per CU
    hipStream_t st = (hipStream_t)stream;
    // row tiles per block.  Latency regime (few blocks per replica): a block's time is its hand-over chain, 32-row blocks give the chain
    // more blocks to overlap.  Throughput regime: a role's busy time per block is ~8.6 k cycles of latency + 6.1 k of MFMA per 32 rows, so
    // 64-row blocks halve the latency share per row
    const int mt = d->B <= 16 ? 1 : (d->B >= FN_DECODE_MT4_ROWS ? 4 : 2);
    const int nblk = (d->B + mt * 16 - 1) / (mt * 16);
    if (nblk > MAXBLK) return FN_E_SHAPE;
    int nrep = prop.multiProcessorCount / per_rep;                     // replicas of the role set that fit one workgroup per CU
    nrep = nrep < nblk ? nrep : nblk;
    const int grid = nrep * per_rep;
    const size_t bp = (size_t)nblk * mt * 16, vp = (size_t)nvt * 16;
    DArgs a;
    a.B = d->B; a.steps = d->steps; a.H = d->H; a.V = d->V; a.nvt = nvt; a.nblk = nblk; a.nrep = nrep;
    a.start_token = d->start_token; a.tok_ld = d->tok_ld;
    a.w1 = d->w_hh1_frag; a.bhh1 = d->b_hh1; a.bih1 = d->b_ih1; a.table1 = d->table1; a.rowbias1 = d->rowbias1; a.h0 = d->h0;
    a.wi2 = d->w_ih2_frag; a.bih2 = d->b_ih2; a.w2 = d->w_hh2_frag; a.bhh2 = d->b_hh2;
    a.wo = d->w_out_frag; a.bo = d->b_out;
    a.x1 = d->ws; a.x2 = a.x1 + 2 * bp * d->H; a.g2 = a.x2 + 2 * bp * d->H; a.logits = a.g2 + bp * 3 * d->H;
    (void)vp;
    a.tokens = d->tokens; a.logp = d->logp;
    a.sync = reinterpret_cast<u32*>(d->sync_ws);
    // h0 -> slot 1 of the layer-1 exchange (the operand of step 0), rows padded with zeros
    int rc = launch_pack(d->h0, d->B, d->H, d->H, a.x1 + bp * d->H, st);
    if (rc != FN_OK) return rc;
    hipError_t me = hipMemsetAsync(a.sync, 0, (size_t)ERRW * 4, st);             // counters only: the error word is sticky
    if (me != hipSuccess) return (int)me;
    return mt == 1 ? launch_decode<1>(a, grid, st) : (mt == 2 ? launch_decode<2>(a, grid, st) : launch_decode<4>(a, grid, st));
}

}  // extern "C"
