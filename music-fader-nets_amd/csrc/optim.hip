// optim.hip - clip_grad_norm_(params, max_norm) + torch.optim.Adam defaults over one flat fp32 buffer
// (trainer_gmm.py:250-251, optimizer built at trainer_gmm.py:52).  HBM-bound elementwise work.
#include "common.h"

namespace {

constexpr int SS_BLOCKS = 1024;

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n4; i += (long)gridDim.x * 256L) {
        const float4 v = g4[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    s = fn_wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(1024) void sumsq_final_kernel(const float* __restrict__ partial, int nb, float* __restrict__ out) {
    __shared__ double red[16];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 1024) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += red[w];
        out[0] = (float)t;
    }
}

// One thread: advances the device-resident counters and derives everything that depends on the step number, so that a
// captured hipGraph of the whole training step never bakes a step-dependent scalar into a kernel argument.
//   counters[0] = training step (beta annealing, trainer_gmm.py:125-128), counters[1] = Adam's t
//   out[0..2] = w_lat, w_cls, w_clf of fn_latent_bwd ; out[3] = lr / (1 - beta1^t) ; out[4] = 1 / sqrt(1 - beta2^t) ; out[5] = beta0 ;
//   out[6] = adversarial weight of the Fader sibling ; out[7] = beta / Bg
__global__ void step_params_kernel(long long* __restrict__ counters, float beta, float lr, float beta1, float beta2, int supervised,
                                   float inv_bg, int advance, float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const long long step = counters[0];
    const long long t = counters[1] + (advance ? 1 : 0);
    double beta0 = 0.0;
    if (step >= 1000) {
        const double a = (double)(step - 10000) / 10000.0 * (double)beta;
        beta0 = a < (double)beta ? a : (double)beta;
    }
    out[0] = (float)(beta0 * inv_bg);
    out[1] = supervised ? 0.0f : (float)(beta0 * inv_bg);
    out[2] = supervised ? inv_bg : 0.0f;
    const double tt = (double)(t > 0 ? t : 1);
    out[3] = (float)((double)lr / (1.0 - pow((double)beta1, tt)));
    out[4] = (float)(1.0 / sqrt(1.0 - pow((double)beta2, tt)));
    out[5] = (float)beta0;
    // sibling trainers: Fader-Networks adversarial weight min(step / 2000 * 1e-4, 1e-4) (trainer_fader.py:105) and the CONSTANT
    // KL weight beta / Bg of the single-encoder VAE, whose loss ignores its own annealed beta0 (trainer_singlevae.py:84-104)
    {
        const double l = (double)step / 2000.0 * 1e-4;
        out[6] = (float)(l < 1e-4 ? l : 1e-4);
    }
    out[7] = beta * inv_bg;
    if (advance) {
        counters[0] = step + 1;
        counters[1] = t;
    }
}

__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, long n, const float* __restrict__ sumsq, float max_norm,
                                                        const float* __restrict__ hyper, float beta1, float beta2, float eps) {
    const float step_size = hyper[0], inv_sqrt_bc2 = hyper[1];
    const float total = sqrtf(sumsq[0]);
    const float coef = fminf(1.0f, max_norm / (total + 1e-6f));      // torch clip_grad_norm_: clamp(max_norm/(norm+1e-6), max=1)
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
        const float gi = g[i] * coef;
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) * inv_sqrt_bc2 + eps;
        p[i] -= step_size * mi / denom;
    }
}

}  // namespace

extern "C" {

size_t fn_sumsq_ws_bytes(int64_t n) { (void)n; return SS_BLOCKS * sizeof(float); }

int fn_sumsq_f32(const float* g, int64_t n, float* out, float* ws, size_t ws_bytes, void* stream) {
    if (!g || !out || !ws) return FN_E_NULL;
    if (n <= 0) return FN_E_SHAPE;
    if ((((uintptr_t)g) & 15) != 0) return FN_E_ALIGN;
    if (ws_bytes < fn_sumsq_ws_bytes(n)) return FN_E_WORKSPACE;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SS_BLOCKS), dim3(256), 0, (hipStream_t)stream, g, (long)n, ws);
    FN_CHECK_LAUNCH();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ws, SS_BLOCKS, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_step_params(int64_t* counters, float beta, float lr, float beta1, float beta2, int supervised, float inv_global_batch,
                   int advance, float* out, void* stream) {
    if (!counters || !out) return FN_E_NULL;
    hipLaunchKernelGGL(step_params_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counters, beta, lr, beta1, beta2,
                       supervised, inv_global_batch, advance, out);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

int fn_clip_adam(float* p, const float* g, float* m, float* v, int64_t n, const float* sumsq, float max_norm, const float* hyper,
                 float beta1, float beta2, float eps, void* stream) {
    if (!p || !g || !m || !v || !sumsq || !hyper) return FN_E_NULL;
    if (n <= 0) return FN_E_SHAPE;
    const long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(clip_adam_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v,
                       (long)n, sumsq, max_norm, hyper, beta1, beta2, eps);
    FN_CHECK_LAUNCH();
    return FN_OK;
}

}  // extern "C"
