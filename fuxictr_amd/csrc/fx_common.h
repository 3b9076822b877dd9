// fx_common.h — shared host/device helpers for libfxctr (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/fxctr.h"

#define FX_WAVE 64

// ---- error plumbing (no exceptions across the C ABI) -------------------------------------
void fx_set_error(const char* fmt, ...);

#define FX_CHECK_ARG(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            fx_set_error(__VA_ARGS__);       \
            return FX_ERR_INVALID;           \
        }                                    \
    } while (0)

#define FX_CHECK_HIP(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            fx_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                     \
            return FX_ERR_HIP;                                                          \
        }                                                                               \
    } while (0)

#define FX_CHECK_LAUNCH()  FX_CHECK_HIP(hipGetLastError())

static inline hipStream_t fx_hip_stream(fx_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int64_t fx_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Row-vector geometry: a D-float row is handled by G lanes (power of two) holding VEC floats each.
struct FxRowGeom {
    int vec;    // 4, 2 or 1
    int lanes;  // power of two, lanes * vec >= D
};
static inline FxRowGeom fx_row_geom(int D) {
    FxRowGeom g;
    g.vec = (D % 4 == 0) ? 4 : ((D % 2 == 0) ? 2 : 1);
    int need = D / g.vec;
    int l = 1;
    while (l < need) l <<= 1;
    g.lanes = l;
    return g;
}

#ifdef __HIPCC__
// ---- device helpers -------------------------------------------------------------------------
template <int VEC>
struct FxVec;
template <>
struct FxVec<4> {
    using T = float4;
};
template <>
struct FxVec<2> {
    using T = float2;
};
template <>
struct FxVec<1> {
    using T = float;
};

template <int VEC>
__device__ __forceinline__ void fx_load(const float* p, float (&r)[VEC]) {
    if constexpr (VEC == 4) {
        float4 t = *reinterpret_cast<const float4*>(p);
        r[0] = t.x; r[1] = t.y; r[2] = t.z; r[3] = t.w;
    } else if constexpr (VEC == 2) {
        float2 t = *reinterpret_cast<const float2*>(p);
        r[0] = t.x; r[1] = t.y;
    } else {
        r[0] = *p;
    }
}
template <int VEC>
__device__ __forceinline__ void fx_store(float* p, const float (&r)[VEC]) {
    if constexpr (VEC == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1], r[2], r[3]);
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<float2*>(p) = make_float2(r[0], r[1]);
    } else {
        *p = r[0];
    }
}

// full-wave (64 lanes) sum, result in every lane
__device__ __forceinline__ float fx_wave_sum(float x) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_xor(x, off, 64);
    return x;
}

// Block sum for blockDim.x == 256 (4 waves); result valid in thread 0. `red` = 4 floats of LDS.
__device__ __forceinline__ float fx_block_sum_256(float x, float* red) {
    x = fx_wave_sum(x);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = x;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}
#endif  // __HIPCC__
