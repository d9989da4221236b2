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

// Entry points that take a graph handle run on the handle's device whatever the caller's current device is, and leave the
// caller's current device as they found it (a library call must not move later torch allocations to another GPU).
struct TlcDeviceScope {
    int prev = -1;
    bool moved = false;
    hipError_t err = hipSuccess;
    explicit TlcDeviceScope(int device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) {
            err = hipSetDevice(device);
            moved = err == hipSuccess;
        }
    }
    ~TlcDeviceScope() {
        if (moved) (void)hipSetDevice(prev);
    }
    TlcDeviceScope(const TlcDeviceScope&) = delete;
    TlcDeviceScope& operator=(const TlcDeviceScope&) = delete;
};
#define TLC_ON_DEVICE(dev)                                                                                   \
    TlcDeviceScope _tlc_scope(dev);                                                                          \
    if (_tlc_scope.err != hipSuccess) {                                                                      \
        tlc_set_error("%s: cannot select device %d: %s", __func__, (int)(dev), hipGetErrorString(_tlc_scope.err)); \
        return TLC_ERR_HIP;                                                                                  \
    }

// ---- device helpers ---------------------------------------------------------------------------------
#ifdef __HIPCC__
__device__ __forceinline__ int tlc_lane() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ unsigned long long tlc_lanemask_lt() {
    return (1ull << tlc_lane()) - 1ull;
}

// value of lane (l ^ J) for J = 1 .. 32, on the vector ALU: DPP quad permutes / row shifts / row rotate for 1, 2, 4, 8 and
// the gfx950 permlane swaps for 16 and 32.  __shfl_xor compiles to ds_bpermute, which goes through the LDS crossbar: with
// eight wavefronts sorting at once that pipe, not latency, was what the register stages of the sort ran at.
template <int J>
__device__ __forceinline__ unsigned tlc_lane_xor_u32(unsigned v) {
    if (J == 1) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);          // quad_perm [1,0,3,2]
    if (J == 2) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);          // quad_perm [2,3,0,1]
    if (J == 4) {
        const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);                     // row_shl:4 -> banks 0, 2
        return (unsigned)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);                 // row_shr:4 -> banks 1, 3
    }
    if (J == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, false);          // row_ror:8
    if (J == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (tlc_lane() & 16) ? r[0] : r[1];
    }
    const auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
    return (tlc_lane() & 32) ? r[0] : r[1];
}


template <int J>
__device__ __forceinline__ unsigned long long tlc_lane_xor_u64(unsigned long long v) {
    return ((unsigned long long)tlc_lane_xor_u32<J>((unsigned)(v >> 32)) << 32) | (unsigned long long)tlc_lane_xor_u32<J>((unsigned)v);
}
__device__ __forceinline__ int tlc_wave_sum_i32(int v) {
    v += (int)tlc_lane_xor_u32<1>((unsigned)v);
    v += (int)tlc_lane_xor_u32<2>((unsigned)v);
    v += (int)tlc_lane_xor_u32<4>((unsigned)v);
    v += (int)tlc_lane_xor_u32<8>((unsigned)v);
    v += (int)tlc_lane_xor_u32<16>((unsigned)v);
    v += (int)tlc_lane_xor_u32<32>((unsigned)v);
    return v;
}
__device__ __forceinline__ long long tlc_wave_sum_i64(long long v) {
    v += (long long)tlc_lane_xor_u64<1>((unsigned long long)v);
    v += (long long)tlc_lane_xor_u64<2>((unsigned long long)v);
    v += (long long)tlc_lane_xor_u64<4>((unsigned long long)v);
    v += (long long)tlc_lane_xor_u64<8>((unsigned long long)v);
    v += (long long)tlc_lane_xor_u64<16>((unsigned long long)v);
    v += (long long)tlc_lane_xor_u64<32>((unsigned long long)v);
    return v;
}
template <int J>
__device__ __forceinline__ double tlc_lane_xor_f64(double v) {
    return __longlong_as_double((long long)tlc_lane_xor_u64<J>((unsigned long long)__double_as_longlong(v)));
}
#define TLC_WAVE_FOLD_F64(OP)                                             \
    { double t;                                                           \
      t = tlc_lane_xor_f64<1>(v);  v = t OP v ? t : v;                    \
      t = tlc_lane_xor_f64<2>(v);  v = t OP v ? t : v;                    \
      t = tlc_lane_xor_f64<4>(v);  v = t OP v ? t : v;                    \
      t = tlc_lane_xor_f64<8>(v);  v = t OP v ? t : v;                    \
      t = tlc_lane_xor_f64<16>(v); v = t OP v ? t : v;                    \
      t = tlc_lane_xor_f64<32>(v); v = t OP v ? t : v; }
__device__ __forceinline__ double tlc_wave_max_f64(double v) { TLC_WAVE_FOLD_F64(>) return v; }
__device__ __forceinline__ double tlc_wave_min_f64(double v) { TLC_WAVE_FOLD_F64(<) return v; }
// inclusive scan across the wavefront: four row shifts inside each row of 16 lanes, then the two row broadcasts
// (lane 15 of a row into the next row; lane 31 into the upper half), all DPP on the vector ALU
__device__ __forceinline__ int tlc_wave_iscan_i32(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);   // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);   // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xE, true);   // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xC, true);   // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, true);   // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, true);   // row_bcast:31 -> rows 2, 3
    return v;
}
// value of a wavefront-uniform lane
__device__ __forceinline__ int tlc_bcast_i32(int v, int src_lane) { return __builtin_amdgcn_readlane(v, src_lane); }
#endif
