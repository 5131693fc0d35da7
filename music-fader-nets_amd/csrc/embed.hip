// embed.hip - gradient of the one-hot columns of W_ih (autograd of  onehot(x) @ W_ih[:, :V]^T,  gmm_model.py:84,89,109,
// 114,132-133): a segmented sum of the per-step gate gradients by token id.  HBM-bound.
#include "common.h"

namespace {

constexpr int NT = 256;

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
