// common.h - shared host/device helpers for libfadernets_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fadernets.h"

// Launch errors surface as positive hipError_t through the C ABI (never exceptions).
#define FN_CHECK_LAUNCH()                        \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

const char* fn_comm_strerror(int code);      // comm.hip: text of FN_COMM_ERROR_BASE + ncclResult_t

__device__ __forceinline__ float fn_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float fn_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ double fn_wave_sum_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
