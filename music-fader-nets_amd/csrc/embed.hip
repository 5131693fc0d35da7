// embed.hip - gradient of the one-hot columns of W_ih (autograd of  onehot(x) @ W_ih[:, :V]^T,  gmm_model.py:84,89,109,
// 114,132-133): out[v][:] = sum of the per-step gate-gradient rows whose input token was v.  HBM-bound (every dgx row
// is read exactly once, as full contiguous rows) and DETERMINISTIC (no floating-point atomics):
//   1. fn_token_sort: counting sort of the (time, batch) positions of a token matrix by token (per-block histograms -> scans ->
//      stable ranks) into a "sort image" {seg, pstart, order}.  ONE sort per batch serves every scan that consumes the same
//      tokens: the four encoder directions and the decoder's layer 1 differ only in the position -> row map (reverse / shift);
//   2. fn_embed_grad_sorted: every token segment is cut into pieces of EG_PIECE rows; one workgroup sums one piece over all
//      columns with 8 independent 16-byte row loads in flight per thread; several scans per launch (grid.y);
//   3. one workgroup per token adds its pieces in order and writes the table row - or the COLUMN of the transposed table,
//      i.e. straight into dW_ih[:, v].
// Also: fn_time_sum_f32, the sum over time of a [T][M] tensor.
#include "common.h"
#include "mma_core.h"

namespace {

constexpr int EG_BLK = 1024;     // rows per sorting block
constexpr int EG_PIECE = 256;    // rows per partial sum (scratch/ab_eg_piece.sh, us per launch at the benchmark shape: 64 rows 177, 128 rows 144, 256 rows 130, 512 rows 140, 1024 rows 185)
constexpr int EG_NT = 384;       // threads of the piece / final kernels: one float4 column each at N3 = 1536
constexpr int EG_MAX_JOBS = 8;

// hist[blk][v] = positions of block blk with token v; position r = tau * B + b
__global__ __launch_bounds__(256) void eg_hist_kernel(int B, long rows, const int* __restrict__ idx, int idx_ld, int V, int* __restrict__ hist) {
    extern __shared__ int cnt[];
    for (int v = threadIdx.x; v < V; v += 256) cnt[v] = 0;
    __syncthreads();
    const long r0 = (long)blockIdx.x * EG_BLK;
    for (int j = threadIdx.x; j < EG_BLK; j += 256) {
        const long r = r0 + j;
        if (r < rows) {
            const int t = idx[(r % B) * idx_ld + r / B];
            atomicAdd(&cnt[min(max(t, 0), V - 1)], 1);
        }
    }
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 256) hist[(long)blockIdx.x * V + v] = cnt[v];
}

// blkoff[blk][v] = positions with token v in earlier blocks; seg[v], pstart[v] = exclusive scans of counts / piece counts
__global__ __launch_bounds__(1024) void eg_scan_kernel(const int* __restrict__ hist, int nblk, int V, int* __restrict__ blkoff,
                                                       int* __restrict__ seg, int* __restrict__ pstart) {
    __shared__ int tot[1024];
    const int v = threadIdx.x;
    int run = 0;
    if (v < V) {
        int b = 0;
        for (; b + 4 <= nblk; b += 4) {                    // four independent loads in flight
            const int h0 = hist[(long)b * V + v], h1 = hist[(long)(b + 1) * V + v], h2 = hist[(long)(b + 2) * V + v],
                      h3 = hist[(long)(b + 3) * V + v];
            blkoff[(long)b * V + v] = run;
            blkoff[(long)(b + 1) * V + v] = run + h0;
            blkoff[(long)(b + 2) * V + v] = run + h0 + h1;
            blkoff[(long)(b + 3) * V + v] = run + h0 + h1 + h2;
            run += h0 + h1 + h2 + h3;
        }
        for (; b < nblk; ++b) {
            blkoff[(long)b * V + v] = run;
            run += hist[(long)b * V + v];
        }
    }
    tot[v] = v < V ? run : 0;
    __syncthreads();
    if (v == 0) {
        int s = 0, ps = 0;
        for (int u = 0; u < V; ++u) {
            seg[u] = s;
            pstart[u] = ps;
            s += tot[u];
            ps += (tot[u] + EG_PIECE - 1) / EG_PIECE;
        }
        seg[V] = s;
        pstart[V] = ps;
    }
}

// order[seg[v] + blkoff[blk][v] + rank] = position   (rank = earlier positions of the same block with the same token: stable)
__global__ __launch_bounds__(256) void eg_scatter_kernel(int B, long rows, const int* __restrict__ idx, int idx_ld, int V,
                                                         const int* __restrict__ blkoff, const int* __restrict__ seg, int* __restrict__ order) {
    __shared__ __attribute__((aligned(16))) int tok[EG_BLK];
    const long r0 = (long)blockIdx.x * EG_BLK;
    for (int j = threadIdx.x; j < EG_BLK; j += 256) {
        const long r = r0 + j;
        tok[j] = r < rows ? min(max(idx[(r % B) * idx_ld + r / B], 0), V - 1) : -1;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < EG_BLK; j += 256) {
        const int t = tok[j];
        if (t < 0) continue;
        int rank = 0;
        const int j4 = j & ~3;
        const int4* tv = reinterpret_cast<const int4*>(tok);
#pragma unroll 4
        for (int q = 0; q < j4; q += 4) {                 // 16-byte LDS reads, independent iterations (the scalar loop was latency-bound)
            const int4 x = tv[q >> 2];
            rank += (x.x == t) + (x.y == t) + (x.z == t) + (x.w == t);
        }
        for (int q = j4; q < j; ++q) rank += (tok[q] == t);
        order[seg[t] + blkoff[(long)blockIdx.x * V + t] + rank] = (int)(r0 + j);
    }
}

struct EgJob {
    const float* dgx;
    float* out;
    float* partial;
    int out_ld, transposed, reverse, idx_shift, start_token;
};
struct EgArgs {
    EgJob job[EG_MAX_JOBS];
    int B, T, N3, V, npieces_max, nstart;
    const int* seg;
    const int* pstart;
    const int* order;
};

// row of dgx_all that position (tau, b) feeds in this scan, or -1: processing step p consumes idx[b][(reverse ? T-1-p : p) + shift]
__device__ __forceinline__ int eg_row(int pos, int B, int T, int reverse, int shift) {
    const int tau = pos / B, b = pos - tau * B;
    const int p = reverse ? T - 1 - (tau - shift) : tau - shift;
    return (p >= 0 && p < T) ? p * B + b : -1;
}

// one workgroup = one piece (<= EG_PIECE sorted positions of one token, or EG_PIECE start-token rows) x all N3 columns
__global__ __launch_bounds__(EG_NT) void eg_piece_kernel(const EgArgs a) {
    __shared__ __attribute__((aligned(16))) int rows_l[EG_PIECE];
    const EgJob& J = a.job[blockIdx.y];
    const int slot = blockIdx.x, V = a.V, B = a.B, T = a.T, N3 = a.N3;
    int cnt = 0;
    if (slot >= a.npieces_max) {
        // rows whose input is the START token: the steps p with tau(p) < 0  (shift = -1, forward: p = 0)
        if (J.idx_shift >= 0) return;
        const int nb = -J.idx_shift * B;                   // such rows are p in [0, -shift): p * B + b
        const int i0 = (slot - a.npieces_max) * EG_PIECE;
        cnt = min(EG_PIECE, nb - i0);
        if (cnt <= 0) return;
        for (int j = threadIdx.x; j < cnt; j += EG_NT) {
            const int p = (i0 + j) / B, b = (i0 + j) % B;
            rows_l[j] = (J.reverse ? T - 1 - p : p) * B + b;
        }
    } else {
        if (slot >= a.pstart[V]) return;
        int lo = 0, hi = V;                                // token v with pstart[v] <= slot < pstart[v+1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (a.pstart[mid] <= slot) lo = mid; else hi = mid;
        }
        const int v = lo;
        const int i0 = a.seg[v] + (slot - a.pstart[v]) * EG_PIECE;
        cnt = min(a.seg[v + 1], i0 + EG_PIECE) - i0;
        for (int j = threadIdx.x; j < cnt; j += EG_NT) rows_l[j] = eg_row(a.order[i0 + j], B, T, J.reverse, J.idx_shift);
    }
    __syncthreads();
    float* dst = J.partial + (long)slot * N3;
    // pad the row list to a multiple of 8 with "row 0, weight 0" so that every load below is unconditional (a branch per row
    // made the compiler serialise: LDS read -> wait -> branch -> load, eight times per batch, with nothing in flight meanwhile)
    for (int j = cnt + threadIdx.x; j < ((cnt + 7) & ~7); j += EG_NT) rows_l[j] = -1;
    __syncthreads();
    const int cnt8 = (cnt + 7) >> 3;
    for (int c = threadIdx.x * 4; c < N3; c += EG_NT * 4) {
        const float* src = J.dgx + c;
        float4 acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        // 32 rows per iteration: four batches of eight asm loads (mma_core.h) are issued back to back, then added batch by batch
        // behind counted waits (24 / 16 / 8 / 0 loads still in flight).  Nothing is in flight across the loop back-edge: hipcc
        // is free to copy registers there, and a copy of a still-loading asm destination would read garbage.  (hipcc's own
        // s_waitcnt placement drained everything before the first add and serialised LDS read -> branch -> load per row.)
        for (int g8 = 0; g8 < cnt8; g8 += 4) {
            f32x4 x[4][8];
            float w[4][8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gc = min(g8 + q, cnt8 - 1);    // batches past the end re-load the last one with weight 0
                const int4 r0 = *reinterpret_cast<const int4*>(rows_l + gc * 8), r1 = *reinterpret_cast<const int4*>(rows_l + gc * 8 + 4);
                const int rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    w[q][u] = (rr[u] >= 0 && g8 + q < cnt8) ? 1.0f : 0.0f;
                    fn_gld4_asm(x[q][u], src + (long)max(rr[u], 0) * N3);
                }
            }
            auto consume = [&](int q) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc[u].x += w[q][u] * x[q][u][0]; acc[u].y += w[q][u] * x[q][u][1];
                    acc[u].z += w[q][u] * x[q][u][2]; acc[u].w += w[q][u] * x[q][u][3];
                }
            };
            fn_wait_vm<24>(); consume(0);
            fn_wait_vm<16>(); consume(1);
            fn_wait_vm<8>(); consume(2);
            fn_wait_vm<0>(); consume(3);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int u = 0; u < 8; ++u) fn_keep(x[q][u]);
        }
        float4 s;
        s.x = ((acc[0].x + acc[1].x) + (acc[2].x + acc[3].x)) + ((acc[4].x + acc[5].x) + (acc[6].x + acc[7].x));
        s.y = ((acc[0].y + acc[1].y) + (acc[2].y + acc[3].y)) + ((acc[4].y + acc[5].y) + (acc[6].y + acc[7].y));
        s.z = ((acc[0].z + acc[1].z) + (acc[2].z + acc[3].z)) + ((acc[4].z + acc[5].z) + (acc[6].z + acc[7].z));
        s.w = ((acc[0].w + acc[1].w) + (acc[2].w + acc[3].w)) + ((acc[4].w + acc[5].w) + (acc[6].w + acc[7].w));
        *reinterpret_cast<float4*>(dst + c) = s;
    }
}

// one workgroup per token: its pieces in order (+ the start-token pieces) -> out[v][:] or, transposed, out[:][v]
__global__ __launch_bounds__(EG_NT) void eg_final_kernel(const EgArgs a) {
    const EgJob& J = a.job[blockIdx.y];
    const int v = blockIdx.x, N3 = a.N3;
    const int p0 = a.pstart[v], p1 = a.pstart[v + 1];
    const bool start = J.idx_shift < 0 && v == J.start_token;
    for (int c = threadIdx.x * 4; c < N3; c += EG_NT * 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = p0; p < p1; ++p) {
            const float4 x = *reinterpret_cast<const float4*>(J.partial + (long)p * N3 + c);
            s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
        }
        if (start)
            for (int p = 0; p < a.nstart; ++p) {
                const float4 x = *reinterpret_cast<const float4*>(J.partial + (long)(a.npieces_max + p) * N3 + c);
                s.x += x.x; s.y += x.y; s.z += x.z; s.w += x.w;
            }
        if (J.transposed) {
            J.out[(long)c * J.out_ld + v] = s.x;
            J.out[(long)(c + 1) * J.out_ld + v] = s.y;
            J.out[(long)(c + 2) * J.out_ld + v] = s.z;
            J.out[(long)(c + 3) * J.out_ld + v] = s.w;
        } else {
            *reinterpret_cast<float4*>(J.out + (long)v * J.out_ld + c) = s;
        }
    }
}

// out[m] = sum_t X[t][m]
__global__ __launch_bounds__(256) void time_sum_kernel(const float* __restrict__ X, int T, long M, float* __restrict__ out) {
    const long m4 = M >> 2;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < m4; i += (long)gridDim.x * 256L) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
        const float4* p = reinterpret_cast<const float4*>(X) + i;
        int t = 0;
        for (; t + 3 < T; t += 4) {
            const float4 x0 = p[(long)t * m4], x1 = p[(long)(t + 1) * m4], x2 = p[(long)(t + 2) * m4], x3 = p[(long)(t + 3) * m4];
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
            a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
            a2.x += x2.x; a2.y += x2.y; a2.z += x2.z; a2.w += x2.w;
            a3.x += x3.x; a3.y += x3.y; a3.z += x3.z; a3.w += x3.w;
        }
        for (; t < T; ++t) {
            const float4 x0 = p[(long)t * m4];
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
        }
        reinterpret_cast<float4*>(out)[i] = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y),
                                                        (a0.z + a1.z) + (a2.z + a3.z), (a0.w + a1.w) + (a2.w + a3.w));
    }
}

struct SortLayout {
    int nblk;
    size_t off_hist, off_blkoff, total;
};
SortLayout sort_layout(int64_t rows, int V) {
    SortLayout L;
    L.nblk = (int)((rows + EG_BLK - 1) / EG_BLK);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    L.off_hist = o; o += up((size_t)L.nblk * V * 4);
    L.off_blkoff = o; o += up((size_t)L.nblk * V * 4);
    L.total = o;
    return L;
}
inline int eg_npieces_max(int64_t rows, int V) { return (int)(rows / EG_PIECE) + V + 1; }
inline int eg_nstart(int B) { return (B + EG_PIECE - 1) / EG_PIECE; }     // start-token rows of a shift = -1 scan

}  // namespace

extern "C" {

size_t fn_token_sort_ints(int64_t rows, int V) { return (size_t)2 * (V + 1) + 2 + (size_t)rows; }
size_t fn_token_sort_ws_bytes(int64_t rows, int V) { return sort_layout(rows, V).total; }

int fn_token_sort(const int32_t* idx, int B, int T, int idx_ld, int V, int32_t* img, void* ws, size_t ws_bytes, void* stream) {
    if (!idx || !img || !ws) return FN_E_NULL;
    if (B <= 0 || T <= 0 || V <= 0 || V > 1024 || idx_ld < T) return FN_E_SHAPE;
    const int64_t rows = (int64_t)B * T;
    const SortLayout L = sort_layout(rows, V);
    if (ws_bytes < L.total) return FN_E_WORKSPACE;
    if (((uintptr_t)ws) & 15) return FN_E_ALIGN;
    char* w = reinterpret_cast<char*>(ws);
    int* hist = reinterpret_cast<int*>(w + L.off_hist);
    int* blkoff = reinterpret_cast<int*>(w + L.off_blkoff);
    int* seg = img;
    int* pstart = img + (V + 1);
    int* order = img + 2 * (V + 1) + 2;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eg_hist_kernel, dim3(L.nblk), dim3(256), (size_t)V * 4, st, B, (long)rows, idx, idx_ld, V, hist);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_scan_kernel, dim3(1), dim3(1024), 0, st, hist, L.nblk, V, blkoff, seg, pstart);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_scatter_kernel, dim3(L.nblk), dim3(256), 0, st, B, (long)rows, idx, idx_ld, V, blkoff, seg, order);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

size_t fn_embed_grad_sorted_ws_bytes(int64_t rows, int B, int V, int N3, int n_jobs) {
    return (size_t)n_jobs * ((size_t)eg_npieces_max(rows, V) + eg_nstart(B)) * N3 * 4;
}

int fn_embed_grad_sorted(const FnEmbedGrad* jobs, int n_jobs, int B, int T, int N3, int V, const int32_t* img, float* ws, size_t ws_bytes,
                         void* stream) {
    if (!jobs || !img || !ws) return FN_E_NULL;
    if (n_jobs <= 0 || n_jobs > EG_MAX_JOBS) return FN_E_COUNT;
    if (B <= 0 || T <= 0 || N3 <= 0 || (N3 & 3) || V <= 0 || V > 1024) return FN_E_SHAPE;
    const int64_t rows = (int64_t)B * T;
    if (ws_bytes < fn_embed_grad_sorted_ws_bytes(rows, B, V, N3, n_jobs)) return FN_E_WORKSPACE;
    if (((uintptr_t)ws) & 15) return FN_E_ALIGN;
    EgArgs a;
    a.B = B; a.T = T; a.N3 = N3; a.V = V;
    a.npieces_max = eg_npieces_max(rows, V);
    a.nstart = eg_nstart(B);
    a.seg = img; a.pstart = img + (V + 1); a.order = img + 2 * (V + 1) + 2;
    const size_t per_job = ((size_t)a.npieces_max + a.nstart) * N3;
    bool any_shift = false;
    for (int j = 0; j < n_jobs; ++j) {
        const FnEmbedGrad& d = jobs[j];
        if (!d.dgx_all || !d.out) return FN_E_NULL;
        if ((((uintptr_t)d.dgx_all)) & 15) return FN_E_ALIGN;
        if (!d.transposed && ((((uintptr_t)d.out) & 15) || (d.out_ld & 3))) return FN_E_ALIGN;
        if (d.idx_shift > 0 || d.idx_shift < -1 || (d.idx_shift && d.reverse) || d.out_ld < (d.transposed ? V : N3)) return FN_E_SHAPE;
        if (d.idx_shift < 0 && (d.start_token < 0 || d.start_token >= V)) return FN_E_SHAPE;
        any_shift |= d.idx_shift < 0;
        EgJob& J = a.job[j];
        J.dgx = d.dgx_all; J.out = d.out; J.partial = ws + (size_t)j * per_job;
        J.out_ld = d.out_ld; J.transposed = d.transposed; J.reverse = d.reverse; J.idx_shift = d.idx_shift; J.start_token = d.start_token;
    }
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eg_piece_kernel, dim3(a.npieces_max + (any_shift ? a.nstart : 0), n_jobs), dim3(EG_NT), 0, st, a);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_final_kernel, dim3(V, n_jobs), dim3(EG_NT), 0, st, a);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

// one-call form (own sort): ws = [sort scratch | sort image | partial sums]
static size_t eg_up(size_t x) { return (x + 255) / 256 * 256; }
size_t fn_embed_grad_ws_bytes(int64_t rows, int V, int N3) {
    // B is not known here: the start-token pieces are bounded by rows / EG_PIECE + 1
    return eg_up(fn_token_sort_ws_bytes(rows, V)) + eg_up(fn_token_sort_ints(rows, V) * 4) +
           ((size_t)eg_npieces_max(rows, V) + (size_t)(rows / EG_PIECE) + 1) * N3 * 4;
}

int fn_embed_grad_f32(const float* dgx_all, int B, int T, int N3, const int32_t* idx, int idx_ld, int idx_shift, int start_token,
                      int reverse, int V, float* out, float* ws, size_t ws_bytes, void* stream) {
    if (!dgx_all || !idx || !out || !ws) return FN_E_NULL;
    if (B <= 0 || T <= 0 || N3 <= 0 || (N3 & 3) || V <= 0 || V > 1024) return FN_E_SHAPE;
    const int64_t rows = (int64_t)B * T;
    if (ws_bytes < fn_embed_grad_ws_bytes(rows, V, N3)) return FN_E_WORKSPACE;
    char* w = reinterpret_cast<char*>(ws);
    const size_t o_img = eg_up(fn_token_sort_ws_bytes(rows, V)), o_part = o_img + eg_up(fn_token_sort_ints(rows, V) * 4);
    int32_t* img = reinterpret_cast<int32_t*>(w + o_img);
    int rc = fn_token_sort(idx, B, T, idx_ld, V, img, w, o_img, stream);
    if (rc != FN_OK) return rc;
    FnEmbedGrad job;
    job.dgx_all = dgx_all; job.out = out; job.out_ld = N3; job.transposed = 0;
    job.reverse = reverse; job.idx_shift = idx_shift; job.start_token = start_token;
    return fn_embed_grad_sorted(&job, 1, B, T, N3, V, img, reinterpret_cast<float*>(w + o_part), ws_bytes - o_part, stream);
}

int fn_time_sum_f32(const float* X, int T, int64_t M, float* out, void* stream) {
    if (!X || !out) return FN_E_NULL;
    if (T <= 0 || M <= 0 || (M & 3)) return FN_E_SHAPE;
    if ((((uintptr_t)X) | ((uintptr_t)out)) & 15) return FN_E_ALIGN;
    const long blocks = (M / 4 + 255) / 256;
    hipLaunchKernelGGL(time_sum_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, X, T, (long)M, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
