// round 5: do vector memory operations retire IN ORDER on vmcnt (gfx950)?  The generated K loops (gen_kloop2/3/4.py) count a YOUNGER store / load as "may stay
// outstanding" when they wait for an OLDER load (`s_waitcnt vmcnt(n)`, n = operations issued behind it) - hipcc does the same.  Here every wave issues an older load
// that has to come from far away (a 4 GB buffer walked with a large stride: L2 / MALL misses) and, right behind it, a younger operation that is fast (a store to /
// a load from a line the wave keeps hot), waits with vmcnt(1) and looks whether the older load's register still holds the poison it was given.
//   hipcc --offload-arch=gfx950 -O3 vmcnt_order.hip -o vmcnt_order && ./vmcnt_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned u32;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE>      // 0: younger sc1 store   1: younger sc1 load (hot line)   2: younger plain load (hot line)   3: younger plain store
__global__ void k(const u32x4* __restrict__ big, long nbig, u32x4* hot, u32* bad, int iters) {
    const long gt = blockIdx.x * (long)blockDim.x + threadIdx.x;
    u32x4* myhot = hot + gt;
    u32 nbad = 0;
    long idx = (gt * 7919) % nbig;
    for (int it = 0; it < iters; ++it) {
        idx = (idx + 1000003L * 37) % nbig;                      // far apart: a new DRAM page every time
        u32x4 a = {0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu, 0xdeadbeefu};
        u32x4 y = {1u, 2u, 3u, (u32)it};
        const u32x4* pa = big + idx;
        if (MODE == 0)
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %2, %3, off sc1\n\ts_waitcnt vmcnt(1)" : "+v"(a) : "v"(pa), "v"(myhot), "v"(y) : "memory");
        else if (MODE == 3)
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %2, %3, off\n\ts_waitcnt vmcnt(1)" : "+v"(a) : "v"(pa), "v"(myhot), "v"(y) : "memory");
        else if (MODE == 1)
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(1)" : "+v"(a), "+v"(y) : "v"(pa), "v"(myhot) : "memory");
        else
            asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off\n\ts_waitcnt vmcnt(1)" : "+v"(a), "+v"(y) : "v"(pa), "v"(myhot) : "memory");
        // a must be big[idx] now: {idx, ~idx, idx * 3, 42}
        const u32 lo = (u32)idx;
        if (a.x != lo || a.y != ~lo || a.z != lo * 3u || a.w != 42u) ++nbad;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (nbad) atomicAdd(bad, nbad);
}

__global__ void fill(u32x4* big, long n) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const u32 lo = (u32)i;
        big[i] = u32x4{lo, ~lo, lo * 3u, 42u};
    }
}

__global__ void hog(const float4* __restrict__ src, float4* dst, long n, int reps) {      // memory traffic beside the test (latency variance)
    float4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r)
        for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            float4 v = src[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    dst[blockIdx.x * (long)blockDim.x + threadIdx.x] = acc;
}

int main() {
    const long nbig = (4L << 30) / 16;
    u32x4 *big, *hot;
    u32* bad;
    float4 *hsrc, *hdst;
    hipMalloc(&big, nbig * 16);
    hipMalloc(&hot, 1024L * 256 * 16);
    hipMalloc(&bad, 4);
    hipMalloc(&hsrc, 1L << 30);
    hipMalloc(&hdst, 2048L * 256 * 16);
    hipMemset(hsrc, 0, 1L << 30);
    hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, big, nbig);
    hipDeviceSynchronize();
    hipStream_t s1, s2;
    hipStreamCreate(&s1);
    hipStreamCreate(&s2);
    const char* names[4] = {"older sc1 load, younger sc1 store ", "older sc1 load, younger sc1 load  ", "older sc1 load, younger plain load", "older sc1 load, younger plain store"};
    for (int beside = 0; beside < 2; ++beside)
        for (int mode = 0; mode < 4; ++mode) {
            hipMemset(bad, 0, 4);
            if (beside) hipLaunchKernelGGL(hog, dim3(1024), dim3(256), 0, s2, hsrc, hdst, (1L << 30) / 16, 6);
            const int iters = 20000;
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(1024), dim3(256), 0, s1, big, nbig, hot, bad, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(1024), dim3(256), 0, s1, big, nbig, hot, bad, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(1024), dim3(256), 0, s1, big, nbig, hot, bad, iters);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(1024), dim3(256), 0, s1, big, nbig, hot, bad, iters);
            hipDeviceSynchronize();
            u32 h = 0;
            hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
            printf("%s, %s: %u of %ld waits found the older load's register still poisoned\n", names[mode], beside ? "beside a streaming kernel" : "alone                    ", h,
                   1024L * 256 * iters);
        }
    return 0;
}
