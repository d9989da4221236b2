// tlc_common.h -- shared host/device helpers of libtlcgnn_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/tlcgnn.h"

#define TLC_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------------
void tlc_set_error(const char* fmt, ...);

#define TLC_HIP_CHECK(expr)                                                                      \
    do {                                                                                         \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess) {                                                                  \
            tlc_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            return TLC_ERR_HIP;                                                                  \
        }                                                                                        \
    } while (0)

#define TLC_REQUIRE(cond, msg)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            tlc_set_error("%s: %s", __func__, msg);              \
            return TLC_ERR_INVALID_ARG;                          \
        }                                                        \
    } while (0)

// ---- device helpers ---------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int tlc_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ unsigned long long tlc_lanemask_lt() {
    return (1ull << tlc_lane()) - 1ull;
}

__device__ __forceinline__ int tlc_wave_sum_i32(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ long long tlc_wave_sum_i64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double tlc_wave_max_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_xor(v, o, 64);
        v = t > v ? t : v;
    }
    return v;
}
__device__ __forceinline__ double tlc_wave_min_f64(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        double t = __shfl_xor(v, o, 64);
        v = t < v ? t : v;
    }
    return v;
}
// inclusive scan across the wave
__device__ __forceinline__ int tlc_wave_iscan_i32(int v) {
    const int lane = tlc_lane();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ int tlc_bcast_i32(int v, int src_lane) { return __shfl(v, src_lane, 64); }
#endif
