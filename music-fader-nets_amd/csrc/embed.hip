// embed.hip - gradient of the one-hot columns of W_ih (autograd of  onehot(x) @ W_ih[:, :V]^T,  gmm_model.py:84,89,109,
// 114,132-133): out[v][:] = sum of the per-step gate-gradient rows whose input token was v.  HBM-bound (every dgx row
// is read exactly once, as full contiguous rows) and DETERMINISTIC (no floating-point atomics):
//   1. counting sort of the row ids by token (per-block histograms -> scans -> stable ranks),
//   2. every token segment is cut into pieces of EG_PIECE rows; one workgroup sums one piece over all columns,
//   3. one workgroup per token adds its pieces in order.
// Also: fn_time_sum_f32, the sum over time of a [T][M] tensor (per-sequence sums of dgx for the z-conditioning weights
// and the bias gradients).
#include "common.h"

namespace {

constexpr int EG_BLK = 1024;     // rows per sorting block
constexpr int EG_PIECE = 128;    // rows per partial sum

__device__ __forceinline__ int row_token(long r, int B, int T, const int* __restrict__ idx, int idx_ld, int idx_shift,
                                         int start_token, int reverse) {
    const int p = (int)(r / B), b = (int)(r % B);
    const int tau = (reverse ? T - 1 - p : p) + idx_shift;
    return tau < 0 ? start_token : idx[(long)b * idx_ld + tau];
}

// hist[blk][v] = rows of block blk with token v
__global__ __launch_bounds__(256) void eg_hist_kernel(int B, int T, const int* __restrict__ idx, int idx_ld, int idx_shift,
                                                      int start_token, int reverse, int V, int* __restrict__ hist) {
    extern __shared__ int cnt[];
    for (int v = threadIdx.x; v < V; v += 256) cnt[v] = 0;
    __syncthreads();
    const long rows = (long)B * T, r0 = (long)blockIdx.x * EG_BLK;
    for (int j = threadIdx.x; j < EG_BLK; j += 256)
        if (r0 + j < rows) atomicAdd(&cnt[row_token(r0 + j, B, T, idx, idx_ld, idx_shift, start_token, reverse)], 1);
    __syncthreads();
    for (int v = threadIdx.x; v < V; v += 256) hist[(long)blockIdx.x * V + v] = cnt[v];
}

// blkoff[blk][v] = rows with token v in earlier blocks; seg[v], pstart[v] = exclusive scans of counts / piece counts
__global__ __launch_bounds__(1024) void eg_scan_kernel(const int* __restrict__ hist, int nblk, int V, int* __restrict__ blkoff,
                                                       int* __restrict__ seg, int* __restrict__ pstart) {
    __shared__ int tot[1024];
    const int v = threadIdx.x;
    int run = 0;
    if (v < V)
        for (int b = 0; b < nblk; ++b) {
            blkoff[(long)b * V + v] = run;
            run += hist[(long)b * V + v];
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

// order[seg[v] + blkoff[blk][v] + rank] = row   (rank = earlier rows of the same block with the same token: stable)
__global__ __launch_bounds__(256) void eg_scatter_kernel(int B, int T, const int* __restrict__ idx, int idx_ld, int idx_shift,
                                                         int start_token, int reverse, int V, const int* __restrict__ blkoff,
                                                         const int* __restrict__ seg, int* __restrict__ order) {
    __shared__ int tok[EG_BLK];
    const long rows = (long)B * T, r0 = (long)blockIdx.x * EG_BLK;
    for (int j = threadIdx.x; j < EG_BLK; j += 256)
        tok[j] = r0 + j < rows ? row_token(r0 + j, B, T, idx, idx_ld, idx_shift, start_token, reverse) : -1;
    __syncthreads();
    for (int j = threadIdx.x; j < EG_BLK; j += 256) {
        const int t = tok[j];
        if (t < 0) continue;
        int rank = 0;
        for (int q = 0; q < j; ++q) rank += (tok[q] == t);
        order[seg[t] + blkoff[(long)blockIdx.x * V + t] + rank] = (int)(r0 + j);
    }
}

// one workgroup = one piece (<= EG_PIECE sorted rows of one token) x all N3 columns (float4 per thread, grid-stride in columns)
__global__ __launch_bounds__(256) void eg_piece_kernel(const float* __restrict__ dgx, int N3, int V, const int* __restrict__ seg,
                                                       const int* __restrict__ pstart, const int* __restrict__ order,
                                                       float* __restrict__ partial) {
    const int slot = blockIdx.x;
    if (slot >= pstart[V]) return;
    int lo = 0, hi = V;                                   // token v with pstart[v] <= slot < pstart[v+1]
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (pstart[mid] <= slot) lo = mid; else hi = mid;
    }
    const int v = lo;
    const int i0 = seg[v] + (slot - pstart[v]) * EG_PIECE, i1 = min(seg[v + 1], i0 + EG_PIECE);
    for (int c = threadIdx.x * 4; c < N3; c += 1024) {
        float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
        int i = i0;
        for (; i + 1 < i1; i += 2) {                      // two independent row loads in flight
            const float4 x0 = *reinterpret_cast<const float4*>(dgx + (long)order[i] * N3 + c);
            const float4 x1 = *reinterpret_cast<const float4*>(dgx + (long)order[i + 1] * N3 + c);
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
            a1.x += x1.x; a1.y += x1.y; a1.z += x1.z; a1.w += x1.w;
        }
        if (i < i1) {
            const float4 x0 = *reinterpret_cast<const float4*>(dgx + (long)order[i] * N3 + c);
            a0.x += x0.x; a0.y += x0.y; a0.z += x0.z; a0.w += x0.w;
        }
        *reinterpret_cast<float4*>(partial + (long)slot * N3 + c) = make_float4(a0.x + a1.x, a0.y + a1.y, a0.z + a1.z, a0.w + a1.w);
    }
}

__global__ __launch_bounds__(256) void eg_final_kernel(const float* __restrict__ partial, int N3, const int* __restrict__ pstart,
                                                       float* __restrict__ out) {
    const int v = blockIdx.x;
    const int p0 = pstart[v], p1 = pstart[v + 1];
    for (int c = threadIdx.x * 4; c < N3; c += 1024) {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int p = p0; p < p1; ++p) {
            const float4 x = *reinterpret_cast<const float4*>(partial + (long)p * N3 + c);
            a.x += x.x; a.y += x.y; a.z += x.z; a.w += x.w;
        }
        *reinterpret_cast<float4*>(out + (long)v * N3 + c) = a;
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

struct EgLayout {
    int nblk, npieces;
    size_t off_hist, off_blkoff, off_seg, off_pstart, off_order, off_partial, total;
};
EgLayout eg_layout(int64_t rows, int V, int N3) {
    EgLayout L;
    L.nblk = (int)((rows + EG_BLK - 1) / EG_BLK);
    L.npieces = (int)(rows / EG_PIECE) + V + 1;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    size_t o = 0;
    L.off_hist = o; o += up((size_t)L.nblk * V * 4);
    L.off_blkoff = o; o += up((size_t)L.nblk * V * 4);
    L.off_seg = o; o += up((size_t)(V + 1) * 4);
    L.off_pstart = o; o += up((size_t)(V + 1) * 4);
    L.off_order = o; o += up((size_t)rows * 4);
    L.off_partial = o; o += up((size_t)L.npieces * N3 * 4);
    L.total = o;
    return L;
}

}  // namespace

extern "C" {

size_t fn_embed_grad_ws_bytes(int64_t rows, int V, int N3) { return eg_layout(rows, V, N3).total; }

int fn_embed_grad_f32(const float* dgx_all, int B, int T, int N3, const int32_t* idx, int idx_ld, int idx_shift, int start_token,
                      int reverse, int V, float* out, float* ws, size_t ws_bytes, void* stream) {
    if (!dgx_all || !idx || !out || !ws) return FN_E_NULL;
    if (B <= 0 || T <= 0 || N3 <= 0 || (N3 & 3) || V <= 0 || V > 1024) return FN_E_SHAPE;
    if ((((uintptr_t)dgx_all) | ((uintptr_t)out) | ((uintptr_t)ws)) & 15) return FN_E_ALIGN;
    const int64_t rows = (int64_t)B * T;
    const EgLayout L = eg_layout(rows, V, N3);
    if (ws_bytes < L.total) return FN_E_WORKSPACE;
    char* w = reinterpret_cast<char*>(ws);
    int* hist = reinterpret_cast<int*>(w + L.off_hist);
    int* blkoff = reinterpret_cast<int*>(w + L.off_blkoff);
    int* seg = reinterpret_cast<int*>(w + L.off_seg);
    int* pstart = reinterpret_cast<int*>(w + L.off_pstart);
    int* order = reinterpret_cast<int*>(w + L.off_order);
    float* partial = reinterpret_cast<float*>(w + L.off_partial);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(eg_hist_kernel, dim3(L.nblk), dim3(256), (size_t)V * 4, st, B, T, idx, idx_ld, idx_shift, start_token, reverse, V, hist);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_scan_kernel, dim3(1), dim3(1024), 0, st, hist, L.nblk, V, blkoff, seg, pstart);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_scatter_kernel, dim3(L.nblk), dim3(256), 0, st, B, T, idx, idx_ld, idx_shift, start_token, reverse, V, blkoff, seg, order);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_piece_kernel, dim3(L.npieces), dim3(256), 0, st, dgx_all, N3, V, seg, pstart, order, partial);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(eg_final_kernel, dim3(V), dim3(256), 0, st, partial, N3, pstart, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
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
