#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { const float* p[8][12]; int tile0[8]; int n; };
__global__ void k_big(const Big a, float* out) {
    int si = 0;
    for (int k = 1; k < 8; ++k) if (k < a.n && (int)blockIdx.x >= a.tile0[k]) si = k;
    const float* q = a.p[si][threadIdx.x & 7];
    if (threadIdx.x == 0) out[blockIdx.x] = q[0];
}
__global__ void k_small(const float* q, float* out) {
    if (threadIdx.x == 0) out[blockIdx.x] = q[0];
}
__global__ void k_dev(const Big* __restrict__ a, float* out) {
    int si = 0;
    for (int k = 1; k < 8; ++k) if (k < a->n && (int)blockIdx.x >= a->tile0[k]) si = k;
    const float* q = a->p[si][threadIdx.x & 7];
    if (threadIdx.x == 0) out[blockIdx.x] = q[0];
}
int main() {
    float *d, *o; hipMalloc(&d, 4096); hipMalloc(&o, 4096 * 4); hipMemset(d, 0, 4096);
    Big h; for (int i = 0; i < 8; ++i) { for (int j = 0; j < 12; ++j) h.p[i][j] = d + j; h.tile0[i] = i * 64; } h.n = 4;
    Big* dh; hipMalloc(&dh, sizeof(Big)); hipMemcpy(dh, &h, sizeof(Big), hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int N = 200;
    for (int variant = 0; variant < 3; ++variant) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < N; ++i) {
                if (variant == 0) k_big<<<256, 256>>>(h, o);
                else if (variant == 1) k_small<<<256, 256>>>(d, o);
                else k_dev<<<256, 256>>>(dh, o);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.2f us per dependent launch\n", variant == 0 ? "1KB kernarg struct, dynamic index" : variant == 1 ? "tiny kernarg" : "descriptor in device memory", ms * 1e3 / N);
        }
    }
    return 0;
}
