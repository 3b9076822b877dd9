// fx_gemm_int.h — definitions shared by the two GEMM translation units of libfxctr (internal, not part of
// the C ABI): fx_gemm.hip (fp32 MFMA kernels, skinny kernels, dispatch) and fx_gemm_x6.hip (the split-bf16
// kernels).  Everything here used to live at the top of fx_gemm.hip.
#pragma once
#include "fx_common.h"

#include <type_traits>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define FX_BK 32

struct GemmArgs {
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    int64_t M, N, K;
    int64_t k_chunk;
    fx_gemm_epilogue epi;
    float* ws;
    int32_t split_k;
    int32_t tiles_m, tiles_n;
    int32_t edge_plain;          // M / N edge tiles run the unmasked k-loop bodies (fx_gemm_pipe_tile)
#ifdef FX_GEMM_LAB
    unsigned long long* trace;   // scripts/ubench/gemm_lab.hip: 8 words per workgroup (timestamps)
#endif
};

#ifdef FX_GEMM_LAB
inline unsigned long long* fx_gemm_lab_trace = nullptr;   // set by the lab before a traced launch
#define FX_LAB_STAMP(slot)                                                                    \
    do {                                                                                      \
        if (a.trace && threadIdx.x == 0)                                                      \
            a.trace[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define FX_LAB_STAMP(slot) do {} while (0)
#endif

__device__ __forceinline__ float fx_epilogue(const fx_gemm_epilogue& e, float z, int64_t m,
                                             int64_t n) {
    if (e.bias) z += e.bias[n];
    if (e.zout) e.zout[m * e.ldz + n] = z;
    if (e.act == 1) z = fmaxf(z, 0.f);
    if (e.mul) z *= e.mul[m * e.ldmul + n];
    if (e.mask) z = e.mask[m * e.ldmask + n] > 0.f ? z : 0.f;
    if (e.add) z += e.add[m * e.ldadd + n];
    return z;
}

// The same epilogue for 4 adjacent columns n .. n+3 of one row (every operand 16-byte aligned: checked
// by the launcher), in two halves: the operand LOADS of a 32x32 accumulator tile are issued together,
// one tile ahead of the arithmetic and the stores.  (Written as load -> use -> store per vector, each
// of a lane's 16 vectors paid its own memory round trip — the compiler may not move a load above a
// store to memory it cannot prove distinct: +6.7 us on a 128x128 tile with bias + ReLU.)  Element for
// element the operation order of fx_epilogue.
struct FxEpiOps4 {
    float4 bias, mul, mask, add;
};

__device__ __forceinline__ void fx_epi_load4(const fx_gemm_epilogue& e, int64_t m, int64_t n,
                                             FxEpiOps4& o) {
    if (e.bias) o.bias = *reinterpret_cast<const float4*>(e.bias + n);
    if (e.mul) o.mul = *reinterpret_cast<const float4*>(e.mul + m * e.ldmul + n);
    if (e.mask) o.mask = *reinterpret_cast<const float4*>(e.mask + m * e.ldmask + n);
    if (e.add) o.add = *reinterpret_cast<const float4*>(e.add + m * e.ldadd + n);
}

__device__ __forceinline__ float4 fx_epi_apply4(const fx_gemm_epilogue& e, float4 z, int64_t m,
                                                int64_t n, const FxEpiOps4& o) {
    if (e.bias) { z.x += o.bias.x; z.y += o.bias.y; z.z += o.bias.z; z.w += o.bias.w; }
    if (e.zout) *reinterpret_cast<float4*>(e.zout + m * e.ldz + n) = z;
    if (e.act == 1) {
        z.x = fmaxf(z.x, 0.f); z.y = fmaxf(z.y, 0.f); z.z = fmaxf(z.z, 0.f); z.w = fmaxf(z.w, 0.f);
    }
    if (e.mul) { z.x *= o.mul.x; z.y *= o.mul.y; z.z *= o.mul.z; z.w *= o.mul.w; }
    if (e.mask) {
        z.x = o.mask.x > 0.f ? z.x : 0.f; z.y = o.mask.y > 0.f ? z.y : 0.f;
        z.z = o.mask.z > 0.f ? z.z : 0.f; z.w = o.mask.w > 0.f ? z.w : 0.f;
    }
    if (e.add) { z.x += o.add.x; z.y += o.add.y; z.z += o.add.z; z.w += o.add.w; }
    return z;
}

template <int I, int N, typename F>
__device__ __forceinline__ void fx_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        fx_static_for<I + 1, N>(f);
    }
}

#define FX_MULTI_MAX 4
struct MultiArgs {
    GemmArgs p[FX_MULTI_MAX];
    int32_t start[FX_MULTI_MAX + 1];     // first workgroup of each problem
    int32_t cfg[FX_MULTI_MAX];
    int32_t n;
};


// ---- fx_gemm_x6.hip ---------------------------------------------------------------------------------
// fp32-accurate GEMM on the bf16 matrix cores (operands split into three exact bf16 planes inside the
// kernel, six v_mfma_f32_32x32x16_bf16 products per fp32 product, fp32 accumulate), 128x128 tiles,
// 512 threads.  `a` prepared by fx_gemm_prepare with tiles_m / tiles_n counted in 128x128 tiles.
bool fx_gemm_x6_enabled();                      // FX_GEMM_BF16X6 (default 1)
int fx_gemm_x6_launch(bool a_kc, bool b_kc, const GemmArgs& a, hipStream_t s);
// cfg[i] bit 0: A k-contiguous, bit 1: B k-contiguous (bit 2 must be clear: 128x128 tiles only)
int fx_gemm_x6_launch_multi(const MultiArgs& ma, int64_t workgroups, hipStream_t s);
