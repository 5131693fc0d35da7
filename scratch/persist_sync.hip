// feasibility: cost of one "publish my slice -> wait for my group -> read the group's slab" round inside a persistent kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32;
__device__ __forceinline__ void store_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 load_cnt(u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid = G groups x S slices; each WG writes ROWS x 16 floats per round into slab[round][group][ROWS][S*16], then reads the whole
// [ROWS][S*16] slab of its group for this round (float4 loads), accumulating a checksum.
template <int ROWS>
__global__ __launch_bounds__(256) void persist(float* slab, u32* counters, int S, int rounds, float* out, u32* err, int do_read) {
    const int group = blockIdx.x / S, slice = blockIdx.x % S;
    const int W = S * 16;
    float acc = 0.f;
    for (int r = 0; r < rounds; ++r) {
        float* base = slab + ((long)r * gridDim.x / S + group) * ROWS * W;
        // publish: ROWS x 16 values, lane -> (row, unit)
        for (int i = threadIdx.x; i < ROWS * 16; i += 256) {
            const int row = i >> 4, u = i & 15;
            store_sc1(base + (long)row * W + slice * 16 + u, (float)(r + 1) + 0.001f * (slice * 16 + u));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(&counters[group * 32], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u32 target = (u32)S * (r + 1);
            u32 spins = 0;
            while (load_cnt(&counters[group * 32]) < target) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > 2000000u) { err[0] = 1; break; }
            }
        }
        __syncthreads();
        if (do_read) {
            const float4* p4 = reinterpret_cast<const float4*>(base);
            for (int i = threadIdx.x; i < ROWS * W / 4; i += 256) { const float4 v = p4[i]; acc += v.x + v.y + v.z + v.w; }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
int main() {
    const int rounds = 256;
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int ROWS = (cfg & 1) ? 128 : 32;
        const int S = 32, G = (cfg & 1) ? 8 : 8;      // 256 WGs
        const int do_read = cfg < 2 ? 1 : 0;
        const long slabf = (long)rounds * G * ROWS * S * 16;
        float *slab, *out; u32 *cnt, *err;
        hipMalloc(&slab, slabf * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cnt, 4096 * 4); hipMalloc(&err, 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cnt, 0, 4096 * 4); hipMemset(err, 0, 4);
            hipEventRecord(e0);
            if (ROWS == 32) persist<32><<<G * S, 256>>>(slab, cnt, S, rounds, out, err, do_read);
            else persist<128><<<G * S, 256>>>(slab, cnt, S, rounds, out, err, do_read);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            u32 herr; hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost);
            float h0; hipMemcpy(&h0, out, 4, hipMemcpyDeviceToHost);
            if (rep == 2) printf("ROWS=%3d groups=%d x %d slices read=%d: %.2f us per round (err=%u, chk=%g)\n", ROWS, G, S, do_read, ms * 1e3 / rounds, herr, h0);
        }
        hipFree(slab); hipFree(out); hipFree(cnt); hipFree(err);
    }
    return 0;
}
