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

// ---- fx_sort.hip: device-wide stable LSD radix sort of (uint32 key, uint32 value) pairs --------
size_t fx_sort_temp_bytes(int64_t n);
size_t fx_sort_zero_words(int64_t n);   // leading words of `temp` that must be 0 before the sort
int fx_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n,
                      unsigned end_bit, void* temp, bool zeroed, hipStream_t s);

// ---- fx_fused.hip: the column fast path of the de-dup (one in-LDS sort per id column + scan / scatter of
// the unique rows), also used by fx_dedup (fx_sparse.hip).  col_cnt: >= C words, col_scan: >= B*C words.
int fx_dedup_columns_launch(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                            const int64_t* col_row_base, const int32_t* col_vocab,
                            const int32_t* col_pad, uint32_t* col_cnt, uint32_t* col_scan,
                            uint32_t* sorted_key, uint32_t* sorted_pos, uint32_t* uniq_row,
                            uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                            fx_scalars* begin_scal, hipStream_t s);

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
// Opens an optimizer step: t += 1 and torch.optim.Adam's per-step scalars — bias_correction1 =
// 1 - beta1 ** step (python double), step_size = lr / bias_correction1, bias_correction2_sqrt =
// (1 - beta2 ** step) ** 0.5.  One thread of one kernel per step (fx_opt_begin_step, or fused into
// the first launch of the de-dup).
__device__ __forceinline__ void fx_begin_step_dev(fx_scalars* sc) {
    const int t = sc->step + 1;
    sc->step = t;
    const double b1 = (double)sc->beta1, b2 = (double)sc->beta2;
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    sc->bc1 = (float)bc1;
    sc->bc2_sqrt = (float)sqrt(bc2);
    sc->step_size = (float)((double)sc->lr / bc1);
}

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
// ---- bf16 table storage (opt-in `emb_dtype: bf16`): rows are read as bf16 and widened, arithmetic and
// optimizer state stay fp32, results are rounded to nearest-even on the way back -----------------
__device__ __forceinline__ float fx_bf16_to_f32(uint16_t h) {
    return __uint_as_float((uint32_t)h << 16);
}
__device__ __forceinline__ uint16_t fx_f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);                                             // round to nearest even
    return (uint16_t)(u >> 16);
}
// element `off` of a table of fp32 (bf16 == 0) or bf16 (bf16 != 0) values: VEC consecutive elements
template <int VEC>
__device__ __forceinline__ void fx_tab_load(const void* base, int bf16, int64_t off, float (&r)[VEC]) {
    if (!bf16) {
        fx_load<VEC>(reinterpret_cast<const float*>(base) + off, r);
        return;
    }
    const uint16_t* p = reinterpret_cast<const uint16_t*>(base) + off;
    if constexpr (VEC == 4) {
        const uint2 t = *reinterpret_cast<const uint2*>(p);             // one 8-byte load
        r[0] = __uint_as_float(t.x << 16);
        r[1] = __uint_as_float(t.x & 0xffff0000u);
        r[2] = __uint_as_float(t.y << 16);
        r[3] = __uint_as_float(t.y & 0xffff0000u);
    } else if constexpr (VEC == 2) {
        const uint32_t t = *reinterpret_cast<const uint32_t*>(p);
        r[0] = __uint_as_float(t << 16);
        r[1] = __uint_as_float(t & 0xffff0000u);
    } else {
        r[0] = fx_bf16_to_f32(p[0]);
    }
}
template <int VEC>
__device__ __forceinline__ void fx_tab_store(void* base, int bf16, int64_t off, const float (&r)[VEC]) {
    if (!bf16) {
        fx_store<VEC>(reinterpret_cast<float*>(base) + off, r);
        return;
    }
    uint16_t* p = reinterpret_cast<uint16_t*>(base) + off;
    if constexpr (VEC == 4) {
        uint2 t;
        t.x = (uint32_t)fx_f32_to_bf16(r[0]) | ((uint32_t)fx_f32_to_bf16(r[1]) << 16);
        t.y = (uint32_t)fx_f32_to_bf16(r[2]) | ((uint32_t)fx_f32_to_bf16(r[3]) << 16);
        *reinterpret_cast<uint2*>(p) = t;
    } else if constexpr (VEC == 2) {
        *reinterpret_cast<uint32_t*>(p) =
            (uint32_t)fx_f32_to_bf16(r[0]) | ((uint32_t)fx_f32_to_bf16(r[1]) << 16);
    } else {
        p[0] = fx_f32_to_bf16(r[0]);
    }
}

// ---- exact-mode Adam: replay of k missed zero-gradient steps of one row -----------------------------
// A dense torch.optim.Adam moves a row that no batch touches: with g = 0, step j after `last` does
//     m *= beta1 ; v *= beta2 ; p -= lr/(1-beta1^t) * m / (sqrt(v)/sqrt(1-beta2^t) + eps),  t = last + j.
// Its magnitude decays like (beta1/sqrt(beta2))^j ~ 0.9^j: after FX_REPLAY_MAX steps the remaining
// terms are below fp32 resolution of the accumulated update; the tail only decays m and v — and the loop
// leaves as soon as the steps have stopped moving this lane's elements (see below).
// The loop body is the hot spot of the catch-up kernels (a wave runs as long as its coldest row: up
// to 256 iterations), so it carries no sqrt and no true division: sqrt(v_j) = sqrt(v_0) sqrt(beta2)^j
// is advanced by one multiply, the two bias corrections use v_rcp_f32 / v_rsq_f32 (1 ulp), the
// quotient one v_rcp_f32 — ~6 instructions per element and step instead of ~35.  Against the
// step-by-step fp32 sequence of the reference this differs by O(j * 6e-8) relative in terms that
// have decayed by 0.9^j (tests: exact mode == dense Adam to 5e-6 over 330 steps).
#define FX_REPLAY_MAX 256
#define FX_REPLAY_WINDOW 16
template <int VEC>
__device__ __forceinline__ void fx_adam_replay(float (&p)[VEC], float (&m)[VEC], float (&v)[VEC],
                                               int last, int k_steps, const fx_scalars& sc,
                                               double lb1, double lb2) {
    int kk = k_steps < FX_REPLAY_MAX ? k_steps : FX_REPLAY_MAX;
    // (round 4) The steps that cannot move p are not replayed.  A row that never had a gradient (m = 0: most
    // first touches of a large table) does not move at all.  Otherwise |u_j| shrinks by >= 8 % per step once
    // t >= 32 (beta1 / sqrt(beta2), over the ratio of the bias corrections), so after a window of steps in which none of
    // this lane's elements changed none will change again — the reference's own fp32 `p -= u_j` is a no-op from
    // there on; m and v take the remaining decay in closed form, as they do past FX_REPLAY_MAX.  With every
    // touched row 256 steps behind, the loop was 100 us of VALU work per DeepFM step at step 300 (bench.py
    // --warmup 300: 1.068 ms against 0.973 at step 50, profiles/r04_gpu_visit_final_summary.txt).
    bool moving = false;
#pragma unroll
    for (int k = 0; k < VEC; ++k) moving |= (m[k] != 0.f);
    if (!moving) kk = 0;
    int done = kk;
    float pw1 = (float)exp2(lb1 * (double)last);       // beta1^last
    float pw2 = (float)exp2(lb2 * (double)last);
    const float sb2 = sqrtf(sc.beta2);
    float r[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) r[k] = sqrtf(v[k]);
    // (windows of FX_REPLAY_WINDOW steps; the test sits between them, the step loop itself has no exit: p at the
    // window's start is compared once per window — every step moves an element the same way, the sign of m, so
    // "equal after a window" is "never moved".  The window test costs ~25 instructions: 16 steps per window
    // keep it at 5 % of the loop; with 8 the first 100 steps of a run — nobody is far enough behind to leave
    // early — paid 4 us per step for it, profiles/r04_gpu_visit_final4_summary.txt.)
    for (int j = 0; j < kk;) {
        const int jend = j + FX_REPLAY_WINDOW < kk ? j + FX_REPLAY_WINDOW : kk;
        float p0[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) p0[k] = p[k];
        for (; j < jend; ++j) {
            pw1 *= sc.beta1;
            pw2 *= sc.beta2;
            const float ss = sc.lr * __builtin_amdgcn_rcpf(1.f - pw1);     // lr / (1 - beta1^t)
            const float ib = __builtin_amdgcn_rsqf(1.f - pw2);              // 1 / sqrt(1 - beta2^t)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                m[k] *= sc.beta1;
                r[k] *= sb2;
                p[k] = fmaf(-ss * m[k], __builtin_amdgcn_rcpf(fmaf(r[k], ib, sc.eps)), p[k]);
            }
        }
        bool changed = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) changed |= (p[k] != p0[k]);
        if (!changed && last + j > 32 && j < kk) {
            done = j;
            break;
        }
    }
    const float f2 = (float)exp2(lb2 * (double)k_steps);
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] *= f2;
    if (k_steps > done) {
        const float f1 = (float)exp2(lb1 * (double)(k_steps - done));
#pragma unroll
        for (int k = 0; k < VEC; ++k) m[k] *= f1;
    }
}
#endif  // __HIPCC__
