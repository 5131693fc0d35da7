// round 4 probe: the K loop of a weight-stationary GRU step (H = 512, 128-row group, 16-unit slice: 32 slices x 8 groups = 256 workgroups) with PRE-SPLIT operands
// on the bf16 MFMA ("bf16 x 6"): weight slice as bf16 triples in LDS (3 gates x 16 k32-blocks x 3 pieces x 1 KB = 144 KB), recurrent operand as bf16
// triples streamed from an exchange slab in L2 ([row tile][block][piece][64 lanes][16 B]: 384 KB per 128-row group, read by the group's 32 slices on ONE XCD).
// Per wave and step: 2 row tiles x 16 blocks: 96 operand loads (1 KB), 144 LDS fragment reads, 576 MFMAs (16 cycles each = 9.2 k cycles; fp32 MFMA: 24.6 k).
// No hand-over, no epilogue: this measures whether the loop is MFMA-, LDS- or L2-bound when all 256 workgroups run it.  Compare: fp32 loop of the same geometry.
// Operand loads are asm statements with counted s_waitcnt (as in the shipped kernels; with plain loads hipcc drains the queue every block: 16-18 us per step for both).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int X6, int D>
__global__ __launch_bounds__(256) void kloop(const u32x4* __restrict__ slab, const u32x4* __restrict__ wimg, float* out, int steps) {
    extern __shared__ __attribute__((aligned(16))) u32x4 wl[];        // X6: [3][16][3][64]; fp32: [3][32 k16-units][4 k4][64 lanes] floats as u32x4 per unit
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.x % 8, slice = blockIdx.x / 8;            // group g on XCD g
    const int nw = X6 ? 3 * 16 * 3 * 64 : 3 * 32 * 64;               // u32x4 elements of the slice
    for (int i = tid; i < nw; i += 256) wl[i] = wimg[(long)slice * nw + i];
    __syncthreads();
    f32x4 acc[2][3];
    for (int m = 0; m < 2; ++m) for (int q = 0; q < 3; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int t0 = 2 * wave;
    for (int st = 0; st < steps; ++st) {
        if (X6) {
            const u32x4* sp = slab + ((long)(g * 2 + (st & 1)) * 8 * 16 * 3) * 64 + lane;      // two slabs alternate like the ping-pong exchange
            u32x4 a[D][2][3];                                                                    // [set][tile][piece], D blocks in flight
            auto load = [&](int set, int blk) {
                for (int m = 0; m < 2; ++m) for (int p = 0; p < 3; ++p)
                    asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a[set][m][p]) : "v"(sp + (((t0 + m) * 16 + blk) * 3 + p) * 64) : "memory");
            };
            auto mma = [&](int set, int blk) {
                for (int q = 0; q < 3; ++q) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, wl[((q * 16 + blk) * 3 + 0) * 64 + lane]);
                    const bf16x8 bm = __builtin_bit_cast(bf16x8, wl[((q * 16 + blk) * 3 + 1) * 64 + lane]);
                    const bf16x8 bl = __builtin_bit_cast(bf16x8, wl[((q * 16 + blk) * 3 + 2) * 64 + lane]);
                    for (int m = 0; m < 2; ++m) {
                        const bf16x8 ah = __builtin_bit_cast(bf16x8, a[set][m][0]), am = __builtin_bit_cast(bf16x8, a[set][m][1]), al = __builtin_bit_cast(bf16x8, a[set][m][2]);
                        f32x4& c = acc[m][q];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                    }
                }
            };
#pragma unroll
            for (int b = 0; b < D - 1; ++b) load(b, b);
#pragma unroll
            for (int blk = 0; blk < 16; ++blk) {
                if (blk + D - 1 < 16) load((blk + D - 1) % D, blk + D - 1);
                // the 6 loads of block blk have landed when at most 6 * (blocks requested behind it) are outstanding
                const int behind = (blk + D - 1 < 16 ? D - 1 : 15 - blk);
                if (behind >= 7) asm volatile("s_waitcnt vmcnt(42)" ::: "memory");
                else if (behind == 6) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
                else if (behind == 5) asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
                else if (behind == 4) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
                else if (behind == 3) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
                else if (behind == 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                else if (behind == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                mma(blk % D, blk);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            // fp32 geometry: unit = 16 k of one row tile = one 1 KB load (lane: 4 consecutive k of row lane & 15 at k-group lane >> 4), 3 gate fragments from LDS, 12 MFMAs
            const u32x4* sp = slab + ((long)(g * 2 + (st & 1)) * 8 * 32) * 64 + lane;
            u32x4 a[2 * D][2];
            auto load = [&](int set, int u) {
                for (int m = 0; m < 2; ++m) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(a[set][m]) : "v"(sp + ((t0 + m) * 32 + u) * 64) : "memory");
            };
            auto mma = [&](int set, int u) {
                for (int q = 0; q < 3; ++q) {
                    const f32x4 b = __builtin_bit_cast(f32x4, wl[(q * 32 + u) * 64 + lane]);
                    for (int m = 0; m < 2; ++m) {
                        const f32x4 av = __builtin_bit_cast(f32x4, a[set][m]);
                        for (int j = 0; j < 4; ++j) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], b[j], acc[m][q], 0, 0, 0);
                    }
                }
            };
#pragma unroll
            for (int b = 0; b < 2 * D - 1; ++b) load(b, b);
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                if (u + 2 * D - 1 < 32) load((u + 2 * D - 1) % (2 * D), u + 2 * D - 1);
                const int behind = (u + 2 * D - 1 < 32 ? 2 * D - 1 : 31 - u);          // units requested behind u: 2 loads each
                switch (behind) {
                    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
                    case 1: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
                    case 2: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
                    case 3: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
                    case 4: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
                    case 5: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
                    case 6: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
                    default: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
                }
                __builtin_amdgcn_sched_barrier(0);
                mma(u % (2 * D), u);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int q = 0; q < 3; ++q) s += acc[m][q][0] + acc[m][q][1] + acc[m][q][2] + acc[m][q][3];
    out[blockIdx.x * 256 + tid] = s;
}

template <int X6, int D>
void run(const char* name) {
    const size_t slab_bytes = (size_t)8 * 2 * 8 * (X6 ? 16 * 3 : 32) * 1024, w_bytes = (size_t)32 * (X6 ? 144 : 96) * 1024;
    void *slab, *w; float* out;
    (void)hipMalloc(&slab, slab_bytes); (void)hipMalloc(&w, w_bytes); (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMemset(slab, 0x11, slab_bytes); (void)hipMemset(w, 0x22, w_bytes);
    const int lds = (X6 ? 144 : 96) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kloop<X6, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    const int steps = 256;
    kloop<X6, D><<<256, 256, lds>>>((const u32x4*)slab, (const u32x4*)w, out, 4);
    (void)hipDeviceSynchronize();
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(e0);
        kloop<X6, D><<<256, 256, lds>>>((const u32x4*)slab, (const u32x4*)w, out, steps);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double flop = 256.0 * steps * 128 * 48 * 512 * 2;
    printf("%-44s %8.3f ms for %d steps = %6.2f us per step  (%6.1f fp32-equivalent TFLOP/s; encoder forward scan today: 12.9 us per step incl. hand-over and epilogue)\n",
           name, best, steps, best * 1e3 / steps, flop / (best * 1e-3) / 1e12);
    (void)hipFree(slab); (void)hipFree(w); (void)hipFree(out);
}

int main() {
    run<0, 2>("fp32 MFMA K loop, 3 units in flight");
    run<0, 4>("fp32 MFMA K loop, 7 units in flight");
    run<1, 2>("bf16 x 6 K loop, 1 block in flight");
    run<1, 4>("bf16 x 6 K loop, 3 blocks in flight");
    run<1, 6>("bf16 x 6 K loop, 5 blocks in flight");
    run<1, 8>("bf16 x 6 K loop, 7 blocks in flight");
    return 0;
}
