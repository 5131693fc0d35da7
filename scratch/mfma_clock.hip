#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, long long* clk) {
    f32x4 a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
    }
    long long t1 = clock64();
    f32x4 s = a0 + a1 + a2 + a3;
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
int main() {
    float* d; long long* c; hipMalloc(&d, 1024 * 256 * 4); hipMalloc(&c, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {200, 2000, 20000, 200000}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0); mfma_loop<<<256, 256>>>(d, iters, c); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long clk; hipMemcpy(&clk, c, 8, hipMemcpyDeviceToHost);
            double flops = 256.0 * 4 * iters * 4 * (2.0 * 16 * 16 * 4);
            printf("iters %7d: %9.3f us  %7.1f TFLOP/s  clock64 delta %lld  -> %.1f cyc/mfma, counter rate %.0f MHz\n", iters, ms * 1e3, flops / (ms * 1e-3) / 1e12, clk,
                   (double)clk / (4.0 * iters), clk / (ms * 1e3));
        }
    }
    return 0;
}
