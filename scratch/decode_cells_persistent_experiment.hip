// decode_cells.hip - the tokens-only greedy decode of 513 .. 2048 sequences (H = 512) as ONE persistent launch: fn_decode_cells_f32
// (gmm_model.py:119-149 with model.eval(), feedback = first-index argmax :73-80).  Built WITHOUT -amdgpu-mfma-vgpr-form (csrc/Makefile):
// with it hipcc 7.2 crashes in its AGPR-copy rewrite pass on this kernel.
#include <cstdio>
#include <cstdlib>

#include <atomic>
#include <type_traits>

#include "gru_layout.h"

namespace {
constexpr int NT = 256;
}

FN_DEVINL void fn_gld4_sb(f32x4& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
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

FN_DEVINL unsigned long long fn_pack_best(float x, int v, int V) {
    const unsigned b = __float_as_uint(x);
    const unsigned key = b ^ ((b >> 31) ? 0xffffffffu : 0x80000000u);
    return ((unsigned long long)key << 32) | (unsigned)(V - 1 - v);
}

// ---------------------------------------------------------------------------------------------------------------
// The three per-token launches above as ONE persistent launch for the whole tokens-only decode (fn_decode_cells_f32; K = H = 512,
// 513 .. 2048 rows): per token 3 launches cost 8-10 us each beside 27 us of MFMA loops at 800 rows.  Every dependency of the token loop
// stays inside a ROW BLOCK of 64 RT rows (RT = ceil(rows / 512), at most 8 blocks): its 32 workgroups (16 hidden units each) exchange the
// layer states through L2 and meet at three monotonic counters per token.  Workgroup (block = blockIdx & 7, slice = blockIdx >> 3): the 32
// workgroups of a block are dispatched to ONE XCD (checked at run time through XCC_ID: the states are written by plain stores and read
// with L1-bypassing loads, which is only coherent inside one XCD's L2 - a mismatch sets the sticky error word and the host takes the
// per-token path).  Per token and workgroup:
//   layer-1 cell   (gru_cell_wlds_ovl_kernel's loop)  h1[t] slice        arrive C1          (the token is read behind C3 of the previous step, in the epilogue)
//   layer-2 cell   waits C1: x = h1[t]; two K phases   h2[t] slice        arrive C2
//   output layer   waits C2: 16 RT rows x 48 columns   packed argmax words by 64-bit atomic max    arrive C3
// The weight slices follow each other through the LDS image without a gap: every half of 16 k steps is requested one load per thread
// and step while the previous half is multiplied (hh1 | ih2 | hh2 | out | hh1 of the next token ...).  Bounded spins, sticky error word.
struct DcArgs {
    int B, steps, V, start_token;
    const float* h0; const float* rowbias; const float* table;
    const float* w_hh1; const float* b_ih1; const float* b_hh1;
    const float* w_ih2; const float* b_ih2; const float* w_hh2; const float* b_hh2;
    const float* w_out; const float* b_out;
    float* h1; float* h2;                            // [2][B][512] each: states of token parity
    unsigned long long* best;                        // [steps][B], zeroed by the caller
    unsigned* sync;                                  // [8 blocks][128 words]: C1 at +0, C2 at +32, C3 at +64, XCC id + 1 of slice 0 at +96; error word at [1024]
};
constexpr unsigned DC_SPIN_LIMIT = 1u << 22;

FN_DEVINL void fn_gld4_sb_sc1(f32x4& dst, unsigned voff, const float* sbase) {
    asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(dst) : "v"(voff), "s"(sbase) : "memory");
}
// a pointer the compiler has lost the uniformity of (values carried around the token loop with its lane-dependent spin loops) back in SGPRs
FN_DEVINL const float* dc_uniform(const float* p) {
    const unsigned long long v = (unsigned long long)(uintptr_t)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const float*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
}
FN_DEVINL unsigned dc_ld_cnt(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <int RT>
__global__ __launch_bounds__(NT) void decode_cells_persist_kernel(const DcArgs a) {
    constexpr int K = 512, H = 512, NKS = K / 16, HALF = NKS / 2, NF = 12, PF = 2;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x4* wl = reinterpret_cast<f32x4*>(dsm);       // [K / 16][3][64]
    float* tr = dsm + 48 * K;
    const int blk = blockIdx.x & 7, slice = blockIdx.x >> 3;
    if ((long)blk * 64 * RT >= a.B) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int li = lane & 15, lg = lane >> 4;
    const int m0 = blk * 64 * RT + wave * 16 * RT, u0 = slice * 16;
    // output-layer tile of this workgroup: rows [blk 64 RT + (slice >> 3) 16 RT, + 16 RT) x columns [48 (slice & 7), + 48); wave w takes row tile min(w, RT - 1)
    const int mo = blk * 64 * RT + (slice >> 3) * 16 * RT + 16 * (wave < RT ? wave : RT - 1), n0 = 48 * (slice & 7);
    const bool out_wave = wave < RT;
    unsigned* c1 = a.sync + blk * 128;
    unsigned* c2 = c1 + 32;
    unsigned* c3 = c1 + 64;
    unsigned* xw = c1 + 96;
    unsigned* err = a.sync + 1024;
    bool gave_up = false;
    auto wait_counter = [&](const unsigned* c, unsigned target) {     // every wave for itself
        unsigned spins = 0;
        while (!gave_up && dc_ld_cnt(c) < target) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0 && (spins > DC_SPIN_LIMIT || dc_ld_cnt(err) != 0)) {
                __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
            }
        }
    };
    auto arrive = [&](unsigned* c) {                 // every wave's stores / atomics have left, then one count per workgroup
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 15u;
    if (slice == 0 && threadIdx.x == 0) __hip_atomic_store(xw, xcc + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // weight-slice sources of the fill chain: lane offset ((rowbase + wave) ld + 4 lane) floats, item j adds whole rows through a scalar base
    struct WSrc { const float* W; int rowbase; bool out; };
    const WSrc S_hh1{a.w_hh1, u0, false}, S_ih2{a.w_ih2, u0, false}, S_hh2{a.w_hh2, u0, false}, S_out{a.w_out, n0, true};
    auto f_off = [&](const WSrc& s) { return (unsigned)((((long)(s.rowbase + wave) * K) + 4 * lane) * 4); };
    auto f_base = [&](const WSrc& s, int half, int j) {
        // cell slices: item j = gate j >> 2, unit wave + 4 (j & 3); output slice: column n0 + wave + 4 j (columns past V - 1 read row V - 1)
        const int rows = s.out ? min(s.rowbase + wave + 4 * j, a.V - 1) - (s.rowbase + wave) : (j >> 2) * H + 4 * (j & 3);       // 32-bit: stays on the scalar unit
        return dc_uniform(s.W + (rows * K + 256 * half));
    };
    f32x4 fv[NF], fa[PF][RT];
    auto fill_store = [&](int half) {
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const int r = wave + 4 * j, c = 64 * half + lane;
            wl[(c >> 2) * 192 + (r >> 4) * 64 + (c & 3) * 16 + (((r & 15) + 4 * (c & 3) + (c >> 2)) & 15)] = fv[j];
        }
    };
    f32x4 arz[RT][2], anx[RT], anh[RT];
    // one K phase over NTL row tiles of this wave (rows r0 + 16 m): acc_rz / accn += A W^T with the slice `cur` (first half already in fv), requesting `nxt`'s first half
    auto phase = [&](auto NTL, const float* A, int r0, f32x4 (&accn)[RT], const WSrc& cur, const WSrc& nxt) {
        constexpr int ntl = decltype(NTL)::value;
        unsigned oa[ntl];
#pragma unroll
        for (int m = 0; m < ntl; ++m) oa[m] = (unsigned)(((long)min(r0 + 16 * m + li, a.B - 1) * K + 4 * lg) * 4);
        const float* pa = dc_uniform(A);
        asm volatile("" : "+s"(pa));
        auto load = [&](int set) {
#pragma unroll
            for (int m = 0; m < ntl; ++m) fn_gld4_sb_sc1(fa[set][m], oa[m], pa);
            pa += 16;
        };
#pragma unroll
        for (int s = 0; s < PF; ++s) load(s);
#pragma unroll
        for (int j = 0; j < NF; ++j) fn_keep(fv[j]);
        fill_store(0);
        __syncthreads();
        f32x4 bq[2][3];
        auto bread = [&](int buf, int s) {
            const int sl = lg * 16 + ((li + 4 * lg + s) & 15);
            bq[buf][0] = wl[s * 192 + sl];
            bq[buf][1] = wl[s * 192 + 64 + sl];
            bq[buf][2] = wl[s * 192 + 128 + sl];
        };
        bread(0, 0);
        // (opaque per phase: otherwise the 96 scalar bases of a token's fill chain are hoisted out of the token loop and spill)
        WSrc curp = cur, nxtp = nxt;
        curp.W = dc_uniform(cur.W);
        nxtp.W = dc_uniform(nxt.W);
        asm volatile("" : "+s"(curp.W), "+s"(nxtp.W));
        const unsigned fo = f_off(cur), fon = f_off(nxt);
        auto step = [&](const int u) __attribute__((always_inline)) {
            auto fillf = [&](int j) { return (j < HALF ? j : j - HALF) < NF ? 1 : 0; };
            auto ringf = [&](int j) { return j + PF < NKS ? ntl : 0; };
            int allowed = u >= PF ? fillf(u - PF) : (PF - 1 - u) * ntl;
#pragma unroll
            for (int d = 1; d < PF; ++d)
                if (u - d >= 0) allowed += ringf(u - d) + fillf(u - d);
            fn_wait_vm_n(allowed);
            if (u != HALF - 1 && u != NKS - 1) bread((u & 1) ^ 1, u + 1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int m = 0; m < ntl; ++m) {
                    arz[m][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][0][j], arz[m][0], 0, 0, 0);
                    arz[m][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][1][j], arz[m][1], 0, 0, 0);
                    accn[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[u % PF][m][j], bq[u & 1][2][j], accn[m], 0, 0, 0);
                }
            if (u + PF < NKS) load(u % PF);
            if (u < HALF) {
                if (u < NF) fn_gld4_sb(fv[u], fo, f_base(curp, 1, u));
            } else if (u - HALF < NF) {
                fn_gld4_sb(fv[u - HALF], fon, f_base(nxtp, 0, u - HALF));
            }
            __builtin_amdgcn_sched_barrier(0);
        };
#pragma unroll
        for (int u = 0; u < HALF; ++u) step(u);
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
            for (int m = 0; m < ntl; ++m) fn_keep(fa[s][m]);
    };
    using NT_RT = std::integral_constant<int, RT>;
    using NT_1 = std::integral_constant<int, 1>;
    // gate epilogue on (row, 4 units) items (cell_epilogue_items with L1-bypassing state loads): h_out = cell(acc, h_prev rows, token row, row constant)
    const int er = lane >> 2, eu = u0 + 4 * (lane & 3);
    float* tw = tr + wave * 4 * 320;
    auto epilogue = [&](const float* h_prev, float* h_out, const float* b_ih, const float* b_hh, auto FIRST_LAYER, int t) {
        constexpr bool first_layer = decltype(FIRST_LAYER)::value;
        f32x4 bi[3], bh[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            bh[q] = *reinterpret_cast<const f32x4*>(b_hh + q * H + eu);
            bi[q] = b_ih ? *reinterpret_cast<const f32x4*>(b_ih + q * H + eu) : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        f32x4 hv[RT], tv[RT][3], rv[RT][3];
        int rows[RT];
#pragma unroll
        for (int m = 0; m < RT; ++m) {
            rows[m] = m0 + 16 * m + er;
            const int rc = min(rows[m], a.B - 1);
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(hv[m]) : "v"(h_prev + (long)rc * H + eu) : "memory");
            if (first_layer) {
                int tok = a.start_token;
                if (t > 0) {
                    unsigned lo;
                    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(lo) : "v"(a.best + (long)(t - 1) * a.B + rc) : "memory");
                    tok = a.V - 1 - (int)lo;
                }
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    tv[m][q] = *reinterpret_cast<const f32x4*>(a.table + (long)tok * 3 * H + q * H + eu);
                    rv[m][q] = *reinterpret_cast<const f32x4*>(a.rowbias + (long)rc * 3 * H + q * H + eu);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int m = 0; m < RT; ++m) fn_keep(hv[m]);
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
                    if (first_layer) e = (e + tv[m][q][c]) + rv[m][q][c];
                    gi[q] = e;
                }
                const float r = fn_sigmoid((gi[0] + bh[0][c]) + g_r[c]);
                const float z = fn_sigmoid((gi[1] + bh[1][c]) + g_z[c]);
                const float n = fn_tanh((gi[2] + g_nx[c]) + r * (g_nh[c] + bh[2][c]));
                o[c] = (1.0f - z) * n + z * hv[m][c];
            }
            if (rows[m] < a.B) *reinterpret_cast<f32x4*>(h_out + (long)rows[m] * H + eu) = o;
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < RT; ++m) arz[m][0] = arz[m][1] = anx[m] = anh[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
    };
    // the chain starts with the first half of the layer-1 slice
    {
        const unsigned fo = f_off(S_hh1);
#pragma unroll
        for (int j = 0; j < NF; ++j) fv[j] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(f_base(S_hh1, 0, j)) + fo);
    }
    const long BH = (long)a.B * H;
#pragma unroll 1
    for (int t = 0; t < a.steps; ++t) {
        float* h1t = a.h1 + (long)(t & 1) * BH;
        float* h2t = a.h2 + (long)(t & 1) * BH;
        const float* h1p = t ? a.h1 + (long)((t - 1) & 1) * BH : a.h0;
        const float* h2p = t ? a.h2 + (long)((t - 1) & 1) * BH : h1t;          // layer 2 starts from the first layer-1 state (gmm_model.py:134-135)
        // ---- layer 1
        zero_acc();
        phase(NT_RT{}, h1p, m0, anh, S_hh1, S_ih2);
        if (t > 0) wait_counter(c3, 32u * (unsigned)t);                          // the previous token's argmax words of this block
        epilogue(h1p, h1t, a.b_ih1, a.b_hh1, std::integral_constant<bool, true>{}, t);
        arrive(c1);
        if (t == 0) {                                                            // all 32 workgroups of the block on one XCD?
            wait_counter(c1, 32u);
            if (dc_ld_cnt(xw) != xcc + 1u) {
                __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
            }
        }
        // ---- layer 2
        zero_acc();
        wait_counter(c1, 32u * (unsigned)(t + 1));
        phase(NT_RT{}, h1t, m0, anx, S_ih2, S_hh2);
        phase(NT_RT{}, h2p, m0, anh, S_hh2, S_out);
        epilogue(h2p, h2t, a.b_ih2, a.b_hh2, std::integral_constant<bool, false>{}, t);
        arrive(c2);
        // ---- output layer + argmax
        zero_acc();
        wait_counter(c2, 32u * (unsigned)(t + 1));
        phase(NT_1{}, h2t, mo, anh, S_out, S_hh1);
        {
            float bv[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) bv[q] = a.b_out[min(n0 + 16 * q + li, a.V - 1)];
            const f32x4 lg3[3] = {arz[0][0], arz[0][1], anh[0]};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned long long w = 0ull;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const int col = n0 + 16 * q + li;
                    const unsigned long long c = col < a.V ? fn_pack_best(lg3[q][i] + bv[q], col, a.V) : 0ull;
                    w = c > w ? c : w;
                }
#pragma unroll
                for (int d = 1; d < 16; d <<= 1) {
                    const unsigned lo = __shfl_xor((unsigned)(w & 0xffffffffull), d, 64), hi = __shfl_xor((unsigned)(w >> 32), d, 64);
                    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
                    w = o > w ? o : w;
                }
                const int row = mo + 4 * lg + i;
                if (li == 0 && out_wave && row < a.B) atomicMax(a.best + (long)t * a.B + row, w);
            }
        }
        arrive(c3);
    }
#pragma unroll
    for (int j = 0; j < NF; ++j) fn_keep(fv[j]);
}

extern "C" {

size_t fn_decode_cells_sync_bytes(void) { return (size_t)(1024 + 32) * sizeof(unsigned); }

int fn_decode_cells_f32(const FnDecodeCells* d, void* stream) {
    if (!d) return FN_E_NULL;
    if (!d->h0 || !d->rowbias || !d->table || !d->w_hh1 || !d->b_hh1 || !d->w_ih2 || !d->w_hh2 || !d->b_hh2 || !d->w_out || !d->b_out || !d->h1 ||
        !d->h2 || !d->best || !d->sync_ws)
        return FN_E_NULL;
    if (d->H != 512 || d->B <= 512 || d->B > 2048 || d->steps <= 0 || d->V <= 0 || d->V > 384 || d->start_token < 0 || d->start_token >= d->V)
        return FN_E_UNSUPPORTED;
    const void* al[] = {d->h0, d->rowbias, d->table, d->w_hh1, d->b_hh1, d->w_ih2, d->w_hh2, d->b_hh2, d->w_out, d->h1, d->h2, d->b_ih1, d->b_ih2};
    for (const void* p : al)
        if (p && (((uintptr_t)p) & 15)) return FN_E_ALIGN;
    if (((uintptr_t)d->best) & 7) return FN_E_ALIGN;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 32) return FN_E_SHAPE;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 256) return FN_E_UNSUPPORTED;
    DcArgs a;
    a.B = d->B; a.steps = d->steps; a.V = d->V; a.start_token = d->start_token;
    a.h0 = d->h0; a.rowbias = d->rowbias; a.table = d->table;
    a.w_hh1 = d->w_hh1; a.b_ih1 = d->b_ih1; a.b_hh1 = d->b_hh1;
    a.w_ih2 = d->w_ih2; a.b_ih2 = d->b_ih2; a.w_hh2 = d->w_hh2; a.b_hh2 = d->b_hh2;
    a.w_out = d->w_out; a.b_out = d->b_out; a.h1 = d->h1; a.h2 = d->h2;
    a.best = reinterpret_cast<unsigned long long*>(d->best); a.sync = reinterpret_cast<unsigned*>(d->sync_ws);
    const int rt = (d->B + 511) / 512;
    const size_t lds = (size_t)48 * 512 * 4 + 4 * 4 * 320 * 4;
    static std::atomic<bool> attr_set[3][32];
    auto launch = [&](auto kern, int slot) -> int {
        if (!attr_set[slot][dev].load(std::memory_order_acquire)) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return (int)e;
            attr_set[slot][dev].store(true, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(256), dim3(NT), lds, (hipStream_t)stream, a);
        FN_CHECK_LAUNCH();
        return FN_OK;
    };
    if (rt == 2) return launch(decode_cells_persist_kernel<2>, 0);
    if (rt == 3) return launch(decode_cells_persist_kernel<3>, 1);
    return launch(decode_cells_persist_kernel<4>, 2);
}

}  // extern "C"
