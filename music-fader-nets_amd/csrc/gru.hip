// gru.hip - GRU sequence scans for the GM-VAE path (encoders gmm_model.py:84,89; sub-decoders :109,114;
// global decoder cells :131-136) and their backward.
//
// One time step of ALL concurrently running scans is one launch: a fused  h_{p-1} W_hh^T  f32-MFMA GEMM
// whose epilogue applies the gate maths and writes h_p (+ the saved gates).  The step boundary is a
// dependent-kernel boundary (~1.5-1.9 us on MI355X), cheaper than an in-kernel grid barrier, and no
// cross-workgroup visibility protocol is needed.  Workgroup tile: 64 batch rows x 16 hidden units x
// {r,z,n}: each wave owns 16 rows and holds r/z/n pre-activations of the SAME (row, unit) in the same
// lane, so the gate epilogue is lane-local.
#include "common.h"
#include "mma_core.h"

namespace {

constexpr int NT = 256;
constexpr int BK = 32;
constexpr int PF_DEPTH = 3;     // K tiles kept in flight ahead of the MFMAs (the scan steps are latency-bound)

// ---------------------------------------------------------------------------------------------
// forward step
// ---------------------------------------------------------------------------------------------
struct FwdStep {
    const float* w_hh;
    const float* b_hh;
    const float* b_ih;
    const float* h_prev;   // null = zeros
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

struct RowsGate {   // local row r of the 48-row weight tile -> row of W_hh ([3H][H]); always in range (H % 16 == 0)
    int h0, H;
    FN_DEVINL bool valid(int) const { return true; }
    FN_DEVINL long clamped(int r) const { return (long)(r >> 4) * H + h0 + (r & 15); }
    FN_DEVINL bool all_valid(int) const { return true; }
};

// NS = number of scans covered by this launch (a distinct kernel symbol per phase: 4 = the encoder step,
// 3 = decoder layer 1 + both sub-decoders, 1 = a single scan), so per-phase durations show up separately in rocprof.
template <int NS>
__global__ __launch_bounds__(NT) void gru_fwd_step_kernel(const FwdArgs args) {
    using SA = Stage<64, BK, true, NT>;
    using SB = Stage<48, BK, true, NT>;
    __shared__ __attribute__((aligned(16))) float smem[2 * (SA::WORDS + SB::WORDS)];

    const int v = fn_xcd_remap(blockIdx.x, args.total);
    int si = 0;
#pragma unroll
    for (int k = 1; k < NS; ++k)
        if (v >= args.s[k].tile0) si = k;
    const FwdStep& S = args.s[si];
    const int local = v - S.tile0;
    const int tn = local / S.ntm, tm = local % S.ntm;     // row tiles fastest: neighbours share weight rows
    const int m0 = tm * 64, hh0 = tn * 16;
    const int B = S.B, H = S.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    f32x4 acc[1][3];
#pragma unroll
    for (int n = 0; n < 3; ++n) acc[0][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (S.h_prev) {
        const RowsPlain ra{m0, B};
        const RowsGate rb{hh0, H};
        const bool vecA = fn_aligned16(S.h_prev, H), vecB = fn_aligned16(S.w_hh, H);
        const int nk = (H + BK - 1) / BK;
        const float* hp = S.h_prev;
        const float* wp = S.w_hh;
        auto loadA = [&](int k0, SA& st) { st.load(hp, H, ra, k0, H, vecA); };
        auto loadB = [&](int k0, SB& st) { st.load(wp, H, rb, k0, H, vecB); };
        fn_kloop<PF_DEPTH, 1, 3, BK, SA, SB>(smem, nk, loadA, loadB, wave * 16, 0, lane, acc);
    }

    const int jj = hh0 + (lane & 15);
    const float bhr = S.b_hh[jj], bhz = S.b_hh[H + jj], bhn = S.b_hh[2 * H + jj];
    float bir = 0.f, biz = 0.f, bin = 0.f;
    if (S.b_ih) { bir = S.b_ih[jj]; biz = S.b_ih[H + jj]; bin = S.b_ih[2 * H + jj]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = m0 + wave * 16 + (lane >> 4) * 4 + i;
        if (b >= B) continue;
        float gxr = bir, gxz = biz, gxn = bin;
        if (S.gx_dense) {
            const float* row = S.gx_dense + (long)b * 3 * H;
            gxr += row[jj]; gxz += row[H + jj]; gxn += row[2 * H + jj];
        }
        if (S.gx_table) {
            const int tok = S.idx ? S.idx[(long)b * S.idx_ld] : S.tok_const;
            const float* row = S.gx_table + (long)tok * 3 * H;
            gxr += row[jj]; gxz += row[H + jj]; gxn += row[2 * H + jj];
        }
        if (S.gx_rowbias) {
            const float* row = S.gx_rowbias + (long)b * 3 * H;
            gxr += row[jj]; gxz += row[H + jj]; gxn += row[2 * H + jj];
        }
        const float ghr = acc[0][0][i] + bhr, ghz = acc[0][1][i] + bhz, ghn = acc[0][2][i] + bhn;
        const float r = fn_sigmoid(gxr + ghr);
        const float z = fn_sigmoid(gxz + ghz);
        const float n = tanhf(gxn + r * ghn);
        const float hp = S.h_prev ? S.h_prev[(long)b * H + jj] : 0.f;
        const float h = (1.0f - z) * n + z * hp;
        S.h_out[(long)b * H + jj] = h;
        if (S.gates) {
            float* g = S.gates + (long)b * 4 * H;
            g[jj] = r; g[H + jj] = z; g[2 * H + jj] = n; g[3 * H + jj] = ghn;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward step:  dh_{q} = [dgh_{q+1} W_hh] + dh_in + dh_ext ; then the gate backward of step q
// ---------------------------------------------------------------------------------------------
struct BwdStep {
    const float* w_hh_t;   // [H][3H]
    const float* a_rz;     // dgx slab of step q+1 ([B][3H], columns [0,2H) used); null = no GEMM
    const float* a_n;      // dghn slab of step q+1 ([B][H])
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

template <int NS>
__global__ __launch_bounds__(NT) void gru_bwd_step_kernel(const BwdArgs args) {
    using SA = Stage<64, BK, true, NT>;
    using SB = Stage<32, BK, true, NT>;
    __shared__ __attribute__((aligned(16))) float smem[2 * (SA::WORDS + SB::WORDS)];

    const int v = fn_xcd_remap(blockIdx.x, args.total);
    int si = 0;
#pragma unroll
    for (int k = 1; k < NS; ++k)
        if (v >= args.s[k].tile0) si = k;
    const BwdStep& S = args.s[si];
    const int local = v - S.tile0;
    const int tn = local / S.ntm, tm = local % S.ntm;
    const int m0 = tm * 64, n0 = tn * 32;
    const int B = S.B, H = S.H;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;

    f32x4 acc[1][2];
    acc[0][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[0][1] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (S.a_rz) {
        const RowsPlain ra{m0, B}, rb{n0, H};
        const bool vecRZ = fn_aligned16(S.a_rz, 3 * H), vecN = fn_aligned16(S.a_n, H), vecB = fn_aligned16(S.w_hh_t, 3 * H);
        const int K = 3 * H, K2 = 2 * H;
        const int nk = (K + BK - 1) / BK;
        const float* arz = S.a_rz;
        const float* an = S.a_n;
        const float* wt = S.w_hh_t;
        auto loadA = [&](int k0, SA& st) {
            if (k0 < K2) st.load(arz, 3 * H, ra, k0, K2, vecRZ);
            else st.load(an, H, ra, k0 - K2, H, vecN);
        };
        auto loadB = [&](int k0, SB& st) { st.load(wt, K, rb, k0, K, vecB); };
        fn_kloop<PF_DEPTH, 1, 2, BK, SA, SB>(smem, nk, loadA, loadB, wave * 16, 0, lane, acc);
    }

#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
        const int jj = n0 + t2 * 16 + (lane & 15);
        if (jj >= H) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int b = m0 + wave * 16 + (lane >> 4) * 4 + i;
            if (b >= B) continue;
            const long o = (long)b * H + jj;
            float dh = acc[0][t2][i];
            if (S.dh_in) dh += S.dh_in[o];
            if (S.dh_ext) dh += S.dh_ext[o];
            if (!S.gates_q) {
                S.dh0_out[o] = dh;
                continue;
            }
            const float* g = S.gates_q + (long)b * 4 * H;
            const float r = g[jj], z = g[H + jj], n = g[2 * H + jj], hn = g[3 * H + jj];
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
            if (S.rowsum) {
                float* rs = S.rowsum + (long)b * 3 * H;
                rs[jj] += drp; rs[H + jj] += dzp; rs[2 * H + jj] += dnp;
            }
            if (S.rowsum_n) S.rowsum_n[o] += dnp * r;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// gradient of the one-hot columns of W_ih: segmented sum of dgx rows by token id
// ---------------------------------------------------------------------------------------------
constexpr int EG_COLS = 64;
constexpr int EG_ROWS = 4096;

__global__ __launch_bounds__(NT) void embed_grad_partial_kernel(const float* __restrict__ dgx, int B, int T, int N3,
                                                                const int* __restrict__ idx, int idx_ld, int idx_shift,
                                                                int start_token, int reverse, int V, float* __restrict__ ws) {
    extern __shared__ __attribute__((aligned(16))) float tab[];   // [V][EG_COLS]
    const int col0 = blockIdx.x * EG_COLS, c = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long rows = (long)B * T;
    const long r0 = (long)blockIdx.y * EG_ROWS, r1 = min(rows, r0 + (long)EG_ROWS);
    for (int i = threadIdx.x; i < V * EG_COLS; i += NT) tab[i] = 0.f;
    __syncthreads();
    if (col0 + c < N3) {
        for (long r = r0 + w; r < r1; r += 4) {
            const int p = (int)(r / B), b = (int)(r % B);
            const int tau = (reverse ? T - 1 - p : p) + idx_shift;
            const int tok = tau < 0 ? start_token : idx[(long)b * idx_ld + tau];
            atomicAdd(&tab[tok * EG_COLS + c], dgx[r * N3 + col0 + c]);
        }
    }
    __syncthreads();
    if (col0 + c < N3)
        for (int vv = w; vv < V; vv += 4) ws[((long)blockIdx.y * V + vv) * N3 + col0 + c] = tab[vv * EG_COLS + c];
}

__global__ void embed_grad_reduce_kernel(const float* __restrict__ ws, int chunks, long total, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < chunks; ++k) s += ws[k * total + i];
        out[i] = s;
    }
}

}  // namespace

extern "C" {

int fn_gru_seq_fwd(const FnGruFwd* scans, int n_scans, void* stream) {
    if (!scans) return FN_E_NULL;
    if (n_scans <= 0 || n_scans > FN_MAX_SCANS) return FN_E_COUNT;
    int Tmax = 0;
    for (int s = 0; s < n_scans; ++s) {
        const FnGruFwd& d = scans[s];
        if (!d.w_hh || !d.b_hh || !d.h_all) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 16) != 0) return FN_E_SHAPE;
        if (d.gx_table && !d.idx) return FN_E_NULL;
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int p = 0; p < Tmax; ++p) {
        FwdArgs a;
        a.n = 0;
        int tiles = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruFwd& d = scans[s];
            if (p >= d.T) continue;
            FwdStep& f = a.s[a.n++];
            const long BH = (long)d.B * d.H;
            f.w_hh = d.w_hh; f.b_hh = d.b_hh; f.b_ih = d.b_ih;
            f.h_prev = p == 0 ? d.h0 : d.h_all + (p - 1) * BH;
            f.h_out = d.h_all + p * BH;
            f.gates = d.gates ? d.gates + p * 4 * BH : nullptr;
            f.gx_dense = d.gx_dense ? d.gx_dense + p * 3 * BH : nullptr;
            f.gx_table = d.gx_table;
            f.gx_rowbias = d.gx_rowbias;
            const int tau = (d.reverse ? d.T - 1 - p : p) + d.idx_shift;
            f.idx = (d.gx_table && tau >= 0) ? d.idx + tau : nullptr;
            f.idx_ld = d.idx_ld;
            f.tok_const = d.start_token;
            f.B = d.B; f.H = d.H;
            f.ntm = (d.B + 63) / 64;
            f.tile0 = tiles;
            tiles += f.ntm * (d.H / 16);
        }
        a.total = tiles;
        switch (a.n) {
#define FN_CASE(N) case N: hipLaunchKernelGGL(gru_fwd_step_kernel<N>, dim3(tiles), dim3(NT), 0, st, a); break;
            FN_CASE(1) FN_CASE(2) FN_CASE(3) FN_CASE(4) FN_CASE(5) FN_CASE(6) FN_CASE(7) FN_CASE(8)
#undef FN_CASE
        }
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
        if (!d.w_hh_t || !d.h_all || !d.gates || !d.dgx_all || !d.dghn_all || !d.scratch) return FN_E_NULL;
        if (d.B <= 0 || d.T <= 0 || d.H <= 0 || (d.H % 16) != 0) return FN_E_SHAPE;
        Tmax = d.T > Tmax ? d.T : Tmax;
    }
    hipStream_t st = (hipStream_t)stream;
    for (int it = 0; it <= Tmax; ++it) {
        BwdArgs a;
        a.n = 0;
        int tiles = 0;
        for (int s = 0; s < n_scans; ++s) {
            const FnGruBwd& d = scans[s];
            if (it > d.T || (it == d.T && !d.dh0)) continue;
            BwdStep& f = a.s[a.n++];
            const long BH = (long)d.B * d.H;
            const int q = d.T - 1 - it;                    // step whose gate backward runs now (-1 on the last)
            f.w_hh_t = d.w_hh_t;
            if (it > 0) {
                f.a_rz = d.dgx_all + (long)(q + 1) * 3 * BH;
                f.a_n = d.dghn_all + (long)(q + 1) * BH;
                f.dh_in = d.scratch;
            } else {
                f.a_rz = nullptr; f.a_n = nullptr;
                f.dh_in = d.dh_last;
            }
            f.dh_ext = (d.dh_ext && q >= 0) ? d.dh_ext + (long)q * BH : nullptr;
            if (q >= 0) {
                f.gates_q = d.gates + (long)q * 4 * BH;
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
            f.ntm = (d.B + 63) / 64;
            f.tile0 = tiles;
            tiles += f.ntm * ((d.H + 31) / 32);
        }
        if (tiles == 0) continue;
        a.total = tiles;
        switch (a.n) {
#define FN_CASE(N) case N: hipLaunchKernelGGL(gru_bwd_step_kernel<N>, dim3(tiles), dim3(NT), 0, st, a); break;
            FN_CASE(1) FN_CASE(2) FN_CASE(3) FN_CASE(4) FN_CASE(5) FN_CASE(6) FN_CASE(7) FN_CASE(8)
#undef FN_CASE
        }
        FN_CHECK_LAUNCH();
    }
    return FN_OK;
}

static int eg_chunks(int64_t rows) { return (int)((rows + EG_ROWS - 1) / EG_ROWS); }
size_t fn_embed_grad_ws_bytes(int64_t rows, int V, int N3) { return (size_t)eg_chunks(rows) * V * N3 * sizeof(float); }

int fn_embed_grad_f32(const float* dgx_all, int B, int T, int N3, const int32_t* idx, int idx_ld, int idx_shift,
                      int start_token, int reverse, int V, float* out, float* ws, size_t ws_bytes, void* stream) {
    if (!dgx_all || !idx || !out || !ws) return FN_E_NULL;
    if (B <= 0 || T <= 0 || N3 <= 0 || V <= 0 || (size_t)V * EG_COLS * sizeof(float) > 160 * 1024) return FN_E_SHAPE;
    const int64_t rows = (int64_t)B * T;
    if (ws_bytes < fn_embed_grad_ws_bytes(rows, V, N3)) return FN_E_WORKSPACE;
    const int chunks = eg_chunks(rows);
    hipStream_t st = (hipStream_t)stream;
    const size_t sh = (size_t)V * EG_COLS * sizeof(float);
    static bool attr_set = false;   // idempotent, value never changes
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)embed_grad_partial_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL(embed_grad_partial_kernel, dim3((N3 + EG_COLS - 1) / EG_COLS, chunks), dim3(NT), sh, st, dgx_all, B, T, N3,
                       idx, idx_ld, idx_shift, start_token, reverse, V, ws);
    FN_CHECK_LAUNCH();
    const long total = (long)V * N3;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(embed_grad_reduce_kernel, dim3(blocks), dim3(256), 0, st, ws, chunks, total, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
