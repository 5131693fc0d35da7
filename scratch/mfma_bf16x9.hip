// round 4 probe: fp32 products from THREE bf16 pieces on the bf16 MFMA (v_mfma_f32_16x16x32_bf16, 16 x the fp32 MFMA rate).
//   x = hi + mid + lo EXACTLY (truncation splits: hi = top 16 bits, mid = top 16 bits of x - hi, lo = the rest: <= 8 significant bits each),
//   a b = sum_ij a_i b_j: 9 exact partial products accumulated in fp32 ("x9"); without the three smallest ("x6": error ~2^-24 |a||b|).
// Measures (1) the pure MFMA streams, (2) MFMA + the on-the-fly split of both operands (what a GEMM that reads fp32 operands pays),
// (3) the accuracy of a K = 4096 dot-product tile against float64 for fp32-MFMA, x9, x6, x3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pack_hi(float a, float b) {          // (bf16 trunc of a) | (bf16 trunc of b) << 16
    return __builtin_amdgcn_perm(__float_as_uint(b), __float_as_uint(a), 0x07060302u);
}
// 8 fp32 -> three bf16x8 (hi, mid, lo), exact
__device__ __forceinline__ void split8(const float (&x)[8], bf16x8& h, bf16x8& m, bf16x8& l) {
    u32x4 H, M, L;
    float r1[8], r2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) r1[i] = x[i] - __uint_as_float(__float_as_uint(x[i]) & 0xffff0000u);
#pragma unroll
    for (int i = 0; i < 8; ++i) r2[i] = r1[i] - __uint_as_float(__float_as_uint(r1[i]) & 0xffff0000u);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        H[i] = pack_hi(x[2 * i], x[2 * i + 1]);
        M[i] = pack_hi(r1[2 * i], r1[2 * i + 1]);
        L[i] = pack_hi(r2[2 * i], r2[2 * i + 1]);
    }
    h = __builtin_bit_cast(bf16x8, H); m = __builtin_bit_cast(bf16x8, M); l = __builtin_bit_cast(bf16x8, L);
}

// mode 0: fp32 MFMA stream (16x16x4 x 8 = one K-32 block of one tile); 1: nine bf16 MFMAs per block, operands pre-split; 2: six;
// 3: nine + split of A and B in the loop (a wave's share of a 64 x 64 wave tile: 4 A + 4 B blocks per 16 tile-blocks); 4: six + split
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed) {
    f32x4 acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float xa[4][8], xb[4][8];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int i = 0; i < 8; ++i) { xa[m][i] = seed * (1 + m + i + threadIdx.x % 7); xb[m][i] = seed * (2 + m * 3 + i); }
    bf16x8 ah[4], am[4], al[4], bh[4], bm[4], bl[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) { split8(xa[m], ah[m], am[m], al[m]); split8(xb[m], bh[m], bm[m], bl[m]); }
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 3) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { xa[m][i] += seed; xb[m][i] -= seed; }     // stands for freshly loaded operands
                split8(xa[m], ah[m], am[m], al[m]);
                split8(xb[m], bh[m], bm[m], bl[m]);
            }
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                f32x4& c = acc[4 * m + n];
                if (MODE == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) c = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[m][j], xb[n][j], c, 0, 0, 0);
                } else {
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bh[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bl[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[m], bm[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[m], bh[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bm[n], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[m], bh[n], c, 0, 0, 0);
                    if (MODE == 1 || MODE == 3) {
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bl[n], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[m], bm[n], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[m], bl[n], c, 0, 0, 0);
                    }
                }
            }
    }
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// accuracy: one 16 x 16 tile, K = Kd: A [16][Kd], B [16][Kd] (both K-contiguous).  Lane l holds for k-block kb: A[l & 15][kb * 32 + (l >> 4) * 8 .. + 8]
template <int MODE>
__global__ void acc_kernel(const float* A, const float* B, float* C, int Kd) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    for (int kb = 0; kb < Kd / 32; ++kb) {
        float xa[8], xb[8];
        for (int i = 0; i < 8; ++i) { xa[i] = A[r * Kd + kb * 32 + g * 8 + i]; xb[i] = B[r * Kd + kb * 32 + g * 8 + i]; }
        if (MODE == 0) {
            // fp32 MFMA 16x16x4: lane (r, g) supplies k = 4 j' + g of each 4-k step: feed the same 32 k values in 8 steps
            for (int j = 0; j < 8; ++j) {
                const float a = A[r * Kd + kb * 32 + j * 4 + g], b = B[r * Kd + kb * 32 + j * 4 + g];
                c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
            }
        } else {
            bf16x8 ah, am, al, bh, bm, bl;
            split8(xa, ah, am, al); split8(xb, bh, bm, bl);
            if (MODE == 9) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, c, 0, 0, 0);
            }
            if (MODE >= 6) {
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c, 0, 0, 0);
            }
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
            c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
        }
    }
    for (int i = 0; i < 4; ++i) C[(g * 4 + i) * 16 + r] = c[i];           // D[row = 4 g + i][col = r] = sum_k A[row][k] B[col][k]
}

template <int MODE>
double rate(const char* name, double flop_per_iter) {
    float* out; hipMalloc(&out, 1024 * 256 * 4);
    const int iters = 4000;
    rate_kernel<MODE><<<1024, 256>>>(out, 10, 1e-3f);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        rate_kernel<MODE><<<1024, 256>>>(out, iters, 1e-3f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    const double tf = flop_per_iter * iters * 1024 * 4 / (best * 1e-3) / 1e12;
    printf("%-64s %8.3f ms  -> %7.1f fp32-equivalent TFLOP/s\n", name, best, tf);
    hipFree(out);
    return tf;
}

template <int MODE>
void accuracy(const char* name, const std::vector<float>& A, const std::vector<float>& B, const std::vector<double>& ref, int Kd, double scale) {
    float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    acc_kernel<MODE><<<1, 64>>>(dA, dB, dC, Kd);
    std::vector<float> C(256); hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
    double worst = 0, rms = 0;
    for (int i = 0; i < 256; ++i) { const double e = std::fabs(C[i] - ref[i]) / scale; worst = e > worst ? e : worst; rms += e * e; }
    printf("%-40s K = %d: max |err| / (|a|.|b|) = %.3e   rms %.3e\n", name, Kd, worst, std::sqrt(rms / 256));
    hipFree(dA); hipFree(dB); hipFree(dC);
}

int main() {
    const double blk = 16.0 * 2 * 16 * 16 * 32;               // 16 tile-blocks of K = 32 per iteration and wave
    rate<0>("fp32 MFMA (v_mfma_f32_16x16x4_f32 x 8 per block)", blk);
    rate<1>("bf16 x9, operands pre-split", blk);
    rate<2>("bf16 x6, operands pre-split", blk);
    rate<3>("bf16 x9 + split of both operands in the loop (64 x 64 wave tile)", blk);
    rate<4>("bf16 x6 + split of both operands in the loop (64 x 64 wave tile)", blk);
    for (int Kd : {512, 4096, 65536}) {
        std::mt19937 rng(7); std::normal_distribution<float> nd(0.f, 1.f);
        std::vector<float> A(16 * (size_t)Kd), B(16 * (size_t)Kd);
        for (auto& v : A) v = nd(rng);
        for (auto& v : B) v = nd(rng) * 0.3f;
        std::vector<double> ref(256);
        double scale = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            double s = 0, sa = 0;
            for (int k = 0; k < Kd; ++k) { s += (double)A[i * Kd + k] * B[j * Kd + k]; sa += std::fabs((double)A[i * Kd + k] * B[j * Kd + k]); }
            ref[i * 16 + j] = s; scale = sa > scale ? sa : scale;
        }
        accuracy<0>("fp32 MFMA", A, B, ref, Kd, scale);
        accuracy<9>("bf16 x9 (all partial products)", A, B, ref, Kd, scale);
        accuracy<6>("bf16 x6", A, B, ref, Kd, scale);
        accuracy<3>("bf16 x3", A, B, ref, Kd, scale);
    }
    return 0;
}
