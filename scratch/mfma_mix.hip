// What slows a v_mfma_f32_16x16x4_f32 stream below 32 cycles/instruction?  Variants of a 16-MFMA body run by 1 wave per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0)
template <int V>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, long long* clk) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
    __syncthreads();
    f32x4 acc[4] = {{0,0,0,0},{0,0,0,0},{0,0,0,0},{0,0,0,0}};
    f32x4 A[4], B[4];
    for (int i = 0; i < 4; ++i) { A[i] = *(const f32x4*)(in + threadIdx.x * 4 + i * 1024); B[i] = *(const f32x4*)(in + 4096 + threadIdx.x * 4 + i * 1024); }
    const float* gp = in + (blockIdx.x & 63) * 4096 + threadIdx.x * 4;
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const unsigned long long base = (unsigned long long)(in + (blockIdx.x & 63) * 4096);
    i32x4 rsrc;
    rsrc[0] = __builtin_amdgcn_readfirstlane((int)(base & 0xffffffffu));
    rsrc[1] = __builtin_amdgcn_readfirstlane((int)((base >> 32) & 0xffff));
    rsrc[2] = 1 << 20;
    rsrc[3] = 0x00020000;
    const int voff = threadIdx.x * 16;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (V == 0) { MF(A[0][0], B[0][0], acc[0]); MF(A[0][0], B[0][0], acc[1]); MF(A[0][0], B[0][0], acc[2]); MF(A[0][0], B[0][0], acc[3]); }
            else { MF(A[q][0], B[q][0], acc[0]); MF(A[q][1], B[q][1], acc[1]); MF(A[q][2], B[q][2], acc[2]); MF(A[q][3], B[q][3], acc[3]); }
            if (V == 2) { B[q] = *(const f32x4*)(lds + ((threadIdx.x * 4 + it * 64 + q * 1024) & 4092)); }
            if (V == 3) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(A[q]) : "v"(gp + ((it * 4 + q) & 15) * 256) : "memory"); }
            if (V == 4) { B[q] = *(const f32x4*)(lds + ((threadIdx.x * 4 + it * 64 + q * 1024) & 4092));
                          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(A[q]) : "v"(gp + ((it * 4 + q) & 15) * 256) : "memory"); }
            if (V == 5) { asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(A[(q + 2) & 3]) : "v"(gp + ((it * 4 + q) & 15) * 256) : "memory"); }
            if (V == 6) { B[(q + 2) & 3] = *(const f32x4*)(lds + ((threadIdx.x * 4 + it * 64 + q * 1024) & 4092)); }
            if (V == 8) { asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(A[(q + 2) & 3]) : "v"(voff + ((it * 4 + q) & 15) * 1024), "s"(rsrc) : "memory"); }
            if (V == 9) { asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(A[(q + 2) & 3]) : "v"(voff + ((it * 4 + q) & 15) * 1024), "s"(rsrc) : "memory"); }
            if (V == 7) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(A[(q + 2) & 3]) : "v"(gp + ((it * 4 + q) & 15) * 256) : "memory"); }
            __builtin_amdgcn_sched_barrier(0);
        }
        if ((V == 3 || V == 4 || V == 5 || V >= 7) && (it & 63) == 63) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    long long t1 = clock64();
    f32x4 s = acc[0] + acc[1] + acc[2] + acc[3] + A[0] + A[1] + A[2] + A[3];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}
template <int V> void run(const char* name, float* d, float* in, long long* c) {
    const int iters = 4000;
    for (int rep = 0; rep < 2; ++rep) k<V><<<256, 256>>>(d, in, iters, c);
    hipDeviceSynchronize();
    long long clk; hipMemcpy(&clk, c, 8, hipMemcpyDeviceToHost);
    printf("%-64s %.1f cycles per MFMA\n", name, (double)clk / (16.0 * iters));
}
int main() {
    float *d, *in; long long* c; hipMalloc(&d, 256 * 256 * 4); hipMalloc(&in, 64 * 4096 * 4 + 65536); hipMalloc(&c, 8);
    hipMemset(in, 0, 64 * 4096 * 4 + 65536);
    run<0>("same operand registers", d, in, c);
    run<1>("16 different operand registers", d, in, c);
    run<2>("+ one ds_read_b128 per 4 MFMAs", d, in, c);
    run<3>("+ one global_load_dwordx4 sc1 per 4 MFMAs (drain per 1024)", d, in, c);
    run<4>("+ both", d, in, c);
    run<5>("global_load into the registers used 2 groups ago", d, in, c);
    run<6>("ds_read_b128 into the registers used 2 groups ago", d, in, c);
    run<7>("global_load (no sc1) into the registers used 2 groups ago", d, in, c);
    run<8>("buffer_load offen into the registers used 2 groups ago", d, in, c);
    run<9>("buffer_load offen sc1 into the registers used 2 groups ago", d, in, c);
    return 0;
}
