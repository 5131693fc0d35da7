#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void probe(unsigned* out) {
    extern __shared__ float smem[];
    smem[threadIdx.x] = 1.f;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);   // HW_ID[15:0]
}
int main() {
    unsigned* d; hipMalloc(&d, 256 * 8 * 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    probe<<<256, 512, 110 * 1024>>>(d);
    unsigned h[256 * 8]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 3; ++b) { printf("block %d:", b); for (int w = 0; w < 8; ++w) printf("  w%d: wave_id %u simd %u cu %u (raw %04x)", w, h[b*8+w] & 15, (h[b*8+w] >> 4) & 3, (h[b*8+w] >> 8) & 15, h[b*8+w]); printf("\n"); }
    return 0;
}
