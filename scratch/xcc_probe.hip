#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(int* out) {
    if (threadIdx.x == 0) {
        int id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        out[blockIdx.x] = id & 0xf;
    }
}
int main() {
    int n = 64; int* d; hipMalloc(&d, n * 4);
    probe<<<n, 256>>>(d);
    int h[64]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) printf("%d ", h[i]);
    printf("\n");
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("CUs %d l2 %d name %s\n", p.multiProcessorCount, p.l2CacheSize, p.name);
    return 0;
}
