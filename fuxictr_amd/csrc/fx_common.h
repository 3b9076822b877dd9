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

// ---- fx_dedup_lds.hip: the bucketed in-LDS de-dup of the generic (n_shards = 1) path: rows hashed into 256
// buckets by their low 8 bits (one stable partition pass), one workgroup sorts a bucket in LDS.  4 launches.
// workspace: fx_dedup_buckets_bytes(n) bytes (<= fx_dedup_workspace_bytes(n)).
size_t fx_dedup_buckets_bytes(int64_t n);
bool fx_dedup_buckets_ok(int64_t n);      // FX_DEDUP_BUCKETS != 0 and n within the 256 x 8192 LDS capacity
int fx_dedup_buckets_launch(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                            const int64_t* col_row_base, const int32_t* col_vocab, const int32_t* col_pad,
                            uint32_t sentinel, void* workspace, uint32_t* sorted_key, uint32_t* sorted_pos,
                            uint32_t* uniq_row, uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
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
// torch.optim.Adam computes its scalars from the python-side doubles (lr 0.001, betas 0.9 / 0.999: the bias
// corrections, step_size = lr / bias_correction1, the lerp weights 1 - beta) while the tensor ops multiply the
// moments by the fp32 images of the betas.  The scalar block holds fp32 images; the double is recovered as
// the nearest number of 6 significant decimal digits when that rounds to the same fp32.  (Round 6, found by
// holding the exact mode to an fp64 trajectory: 1 - fl32(0.999) = 0.99998713e-3 against 1e-3 — the weight of
// g^2 in v was 1.3e-5 low, every update 6e-6 large; 1 - 0.999^t from the fp32 image is off by the same 1.3e-5
// for every t << 1000.)
__device__ __forceinline__ double fx_dec_f64(float b) {
    const double x = (double)b;
    if (!(x > 0.0) || !(x < 1.0e30)) return x;
    const double scale = pow(10.0, 5.0 - floor(log10(x)));
    const double r = rint(x * scale) / scale;
    return ((float)r == b) ? r : x;
}
__device__ __forceinline__ double fx_beta_f64(float b) { return fx_dec_f64(b); }
// the weight torch hands to lerp_ / addcmul_: float(1 - beta) of the double beta
__device__ __forceinline__ float fx_one_minus(float beta) { return (float)(1.0 - fx_dec_f64(beta)); }
// log2 of both: lb* (fp32 image: the decay of the moments), lc* (the double: the bias corrections)
struct FxLogs {
    double lb1, lb2, lc1, lc2;
};
__device__ __forceinline__ FxLogs fx_logs_of(const fx_scalars& sc) {
    FxLogs l;
    l.lb1 = log2((double)sc.beta1);
    l.lb2 = log2((double)sc.beta2);
    l.lc1 = log2(fx_beta_f64(sc.beta1));
    l.lc2 = log2(fx_beta_f64(sc.beta2));
    return l;
}
__device__ __forceinline__ void fx_begin_step_dev(fx_scalars* sc) {
    const int t = sc->step + 1;
    sc->step = t;
    const double b1 = fx_beta_f64(sc->beta1), b2 = fx_beta_f64(sc->beta2);
    const double bc1 = 1.0 - pow(b1, (double)t);
    const double bc2 = 1.0 - pow(b2, (double)t);
    sc->bc1 = (float)bc1;
    sc->bc2_sqrt = (float)sqrt(bc2);
    sc->step_size = (float)(fx_dec_f64(sc->lr) / bc1);
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

// ---- the Adam series table (round 6; layout and mathematics: include/fxctr.h, fx_adam_series_build) ----
// The k missed steps of a row last updated at `last` sum to
//     (lr m / sqrt(v)) * ( F(last; c) - (b1/sqrt(b2))^k F(last + k; c b2^(-k/2)) ),   c = eps / sqrt(v)
// with F(t; .) read from the table: per element one sqrt, one division for 1 / sqrt(v), and per table
// segment (one for t >= 128) a division and six fmas — instead of 3 instructions for each of up to 256
// replayed steps.  The arithmetic of an element does not depend on how many elements a lane holds, so every
// kernel that calls this (unique-row catch-up, owner fetch, flush) moves a row to the same bits.
#define FX_SER_HDR 16
#define FX_SER_SEGW 8
#define FX_SER_ENTRYW (8 * FX_SER_SEGW)
struct FxSeries {
    const float* tab;      // first early entry; nullptr: no table (replay)
    int tcap;
};
__device__ __forceinline__ FxSeries fx_series_of(const fx_scalars* scal, const fx_scalars& sc) {
    FxSeries s;
    s.tcap = sc.series_tcap;
    s.tab = s.tcap > FX_SERIES_EARLY ? reinterpret_cast<const float*>(scal) + 16 + FX_SER_HDR : nullptr;
    return s;
}
__device__ __forceinline__ const float* fx_series_entry(const FxSeries& s, int t, int& nseg) {
    t = t < s.tcap ? t : s.tcap - 1;
    t = t > 0 ? t : 0;
    if (t < FX_SERIES_EARLY) {
        const float* e = s.tab + (int64_t)t * FX_SER_ENTRYW;
        nseg = __float_as_int(e[7]);
        return e;
    }
    nseg = 1;
    return s.tab + (int64_t)FX_SERIES_EARLY * FX_SER_ENTRYW + (int64_t)(t - FX_SERIES_EARLY) * FX_SER_SEGW;
}
// out[e] = F(t; c[e]) of the entry `en`
template <int NE>
__device__ __forceinline__ void fx_series_F(const float* en, int nseg, const float (&c)[NE], float (&out)[NE]) {
#pragma unroll
    for (int e = 0; e < NE; ++e) out[e] = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const float4 a = *reinterpret_cast<const float4*>(en + s * FX_SER_SEGW);
        const float4 b = *reinterpret_cast<const float4*>(en + s * FX_SER_SEGW + 4);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const float y = a.x / (a.x + c[e]);
            float q = fmaf(y, b.z, b.y);
            q = fmaf(y, q, b.x);
            q = fmaf(y, q, a.w);
            q = fmaf(y, q, a.z);
            q = fmaf(y * y, q, a.y);
            out[e] = fmaf(y, q, out[e]);
        }
    }
}
// p -= the k zero-gradient steps after `last` (k > FX_SERIES_KDIR); m, v are the row's moments AT `last`
template <int NE>
__device__ __forceinline__ void fx_series_move(float (&p)[NE], const float (&m)[NE], const float (&v)[NE],
                                               int last, int k, const fx_scalars& sc, const FxSeries& ser,
                                               const FxLogs& lg) {
    float c[NE], scale[NE], F0[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const float r = sqrtf(v[e]);
        const bool live = (m[e] != 0.f) && (r > 0.f);
        const float inv = live ? 1.f / r : 0.f;
        c[e] = live ? sc.eps * inv : 1.f;          // (a dead element: any c, scale 0)
        scale[e] = sc.lr * m[e] * inv;
    }
    int ns;
    const float* en = fx_series_entry(ser, last, ns);
    fx_series_F<NE>(en, ns, c, F0);
    if (k < 1024) {                                // (b1/sqrt(b2))^1024 ~ 1e-47: no tail left
        const float rho = (float)exp2((lg.lb1 - 0.5 * lg.lb2) * (double)k);
        const float bm = (float)exp2(-0.5 * lg.lb2 * (double)k);
        float c2[NE], F1[NE];
#pragma unroll
        for (int e = 0; e < NE; ++e) c2[e] = c[e] * bm;
        const float* en2 = fx_series_entry(ser, last + k, ns);
        fx_series_F<NE>(en2, ns, c2, F1);
#pragma unroll
        for (int e = 0; e < NE; ++e) F0[e] = fmaf(-rho, F1[e], F0[e]);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = fmaf(-scale[e], F0[e], p[e]);
}

// the k <= FX_SERIES_KDIR steps that the table does not cover (the difference of two sums would cancel), in
// the same factored form: sum_i w_i / (g_i + c), met with p once.  The bias corrections are carried as
// d = 1 - b^t (d' = (1 - b) + b d): 1 - b2^t is 0.001 t early in a run and b2^t rounded to fp32 first would
// leave it with a relative error of 3e-5 / t.
template <int NE>
__device__ __forceinline__ void fx_short_move(float (&p)[NE], const float (&m)[NE], const float (&v)[NE],
                                              int last, int k, const fx_scalars& sc, const FxLogs& lg) {
    float c[NE], scale[NE], acc[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const float r = sqrtf(v[e]);
        const bool live = (m[e] != 0.f) && (r > 0.f);
        const float inv = live ? 1.f / r : 0.f;
        c[e] = live ? sc.eps * inv : 1.f;
        scale[e] = sc.lr * m[e] * inv;
        acc[e] = 0.f;
    }
    const float sb2 = sqrtf(sc.beta2);
    // d = 1 - B^t of the doubles B; d' = (1 - B) + B d
    const float e1 = (float)(1.0 - exp2(lg.lc1)), e2 = (float)(1.0 - exp2(lg.lc2));
    float d1 = (float)(1.0 - exp2(lg.lc1 * (double)last)), d2 = (float)(1.0 - exp2(lg.lc2 * (double)last));
    float bi = 1.f, hb = 1.f;
    for (int i = 1; i <= k; ++i) {
        bi *= sc.beta1;
        hb *= sb2;
        d1 = fmaf(sc.beta1, d1, e1);
        d2 = fmaf(sc.beta2, d2, e2);
        const float w = bi * __builtin_amdgcn_rcpf(d1);
        const float g = hb * __builtin_amdgcn_rsqf(d2);
#pragma unroll
        for (int e = 0; e < NE; ++e) acc[e] = fmaf(w, __builtin_amdgcn_rcpf(g + c[e]), acc[e]);
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) p[e] = fmaf(-scale[e], acc[e], p[e]);
}

template <int VEC>
__device__ __forceinline__ void fx_adam_replay(float (&p)[VEC], float (&m)[VEC], float (&v)[VEC],
                                               int last, int k_steps, const fx_scalars& sc,
                                               const FxLogs& lg, const FxSeries& ser) {
    const double lb1 = lg.lb1, lb2 = lg.lb2;
    if (ser.tab != nullptr) {
        bool mv = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) mv |= (m[k] != 0.f);
        if (mv) {
            if (k_steps > FX_SERIES_KDIR) fx_series_move<VEC>(p, m, v, last, k_steps, sc, ser, lg);
            else fx_short_move<VEC>(p, m, v, last, k_steps, sc, lg);
        }
        const float f1 = (float)exp2(lb1 * (double)k_steps), f2 = (float)exp2(lb2 * (double)k_steps);
#pragma unroll
        for (int k = 0; k < VEC; ++k) { m[k] *= f1; v[k] *= f2; }
        return;
    }
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
    float pw1 = (float)exp2(lg.lc1 * (double)last);       // beta1^last
    float pw2 = (float)exp2(lg.lc2 * (double)last);
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
