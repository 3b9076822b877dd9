// fx_cin.hip — xDeepFM Compressed Interaction Network layer, fused: the
// [B, F0*Mi, D] outer-product tensor the reference materialises with einsum (400 MB at B=4096,
// compressed_interaction_net.py:70-71) never exists.
//
// Reference: fuxictr/pytorch/layers/interactions/compressed_interaction_net.py:54-76
//   had[b, h*Mi+m, d] = X0[b,h,d] * Xi[b,m,d]                      (einsum "bhd,bmd->bhmd")
//   Xn[b,o,d]         = sum_c W[o,c] had[b,c,d] + bias[o]          (Conv1d, kernel_size 1)
//   pool[b,o]         = sum_d Xn[b,o,d]
// Persistent workgroups (one per CU) keep the layer's W (O x F0*Mi fp32, 97 KB for 16 x 39*39) in
// LDS and walk the samples; a sample's X0 / Xi tiles are staged in LDS too.  All reductions are in a
// fixed order.  fp32 VALU FMAs: the per-sample products are 39x39x16 — far below an MFMA tile.
#include "fx_common.h"

#define FX_CIN_MAX_W_FLOATS (30 * 1024)   // 120 KB of W (or dW) per workgroup
#define FX_CIN_MAX_TILE 4096              // F0*D and Mi*D and O*D staged per sample

struct CinArgs {
    const float* X0; int64_t x0_ld;
    const float* Xi; int64_t xi_ld;
    const float* W;          // [O, C]   C = F0 * Mi
    const float* bias;       // [O]
    float* Xn;               // [B, O, D]
    float* pool; int64_t pool_ld;   // pool[b*pool_ld + o] = sum_d Xn[b,o,d]
    const float* dXn;        // [B, O, D] or null
    const float* dpool; int64_t dpool_ld;
    float* dX0; int64_t dx0_ld;
    float* dXi; int64_t dxi_ld;
    float* partial;          // [G][O*C + O]
    int64_t B;
    int32_t F0, Mi, D, O, acc_dx0;
};

// ---- forward --------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cin_fwd(CinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.F0 * a.Mi, D = a.D, O = a.O;
    float* Ws = smem;                    // [O*C]
    float* x0 = Ws + O * C;              // [F0*D]
    float* xi = x0 + a.F0 * D;           // [Mi*D]
    for (int t = threadIdx.x; t < O * C; t += 256) Ws[t] = a.W[t];
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.F0 * D; t += 256) x0[t] = a.X0[b * a.x0_ld + t];
        for (int t = threadIdx.x; t < a.Mi * D; t += 256) xi[t] = a.Xi[b * a.xi_ld + t];
        __syncthreads();
        for (int od = threadIdx.x; od < O * D; od += 256) {
            const int o = od / D, d = od - o * D;
            const float* w = Ws + o * C;
            float acc = 0.f;
            for (int h = 0; h < a.F0; ++h) {
                const float xh = x0[h * D + d];
                float s = 0.f;
                for (int m = 0; m < a.Mi; ++m) s = fmaf(w[h * a.Mi + m], xi[m * D + d], s);
                acc = fmaf(xh, s, acc);
            }
            acc += a.bias[o];
            a.Xn[(b * O + o) * D + d] = acc;
        }
        if (a.pool) {
            __syncthreads();   // reuse x0 as scratch? no: read Xn back from global (L2) per o
            for (int o = threadIdx.x; o < O; o += 256) {
                float s = 0.f;
                for (int d = 0; d < D; ++d) s += a.Xn[(b * O + o) * D + d];
                a.pool[b * a.pool_ld + o] = s;
            }
        }
    }
}

// ---- backward: input gradients ------------------------------------------------------------------
// g[o,d] = dXn[b,o,d] + dpool[b,o];  T[h,m,d] = sum_o g[o,d] W[o,h*Mi+m]
// dX0[h,d] = sum_m T Xi[m,d] ; dXi[m,d] = sum_h T X0[h,d]
// thread = (d, hg): hg walks h = hg, hg+NH, ...; dXi partials of the NH h-groups meet in LDS.
__global__ __launch_bounds__(256) void k_cin_bwd_dx(CinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.F0 * a.Mi, D = a.D, O = a.O;
    int Dp = 1;
    while (Dp < D) Dp <<= 1;
    const int NH = 256 / Dp;             // h-groups
    float* Ws = smem;                    // [O*C]
    float* x0 = Ws + O * C;              // [F0*D]
    float* xi = x0 + a.F0 * D;           // [Mi*D]
    float* g = xi + a.Mi * D;            // [O*D]
    float* red = g + O * D;              // [NH][Mi*D]
    for (int t = threadIdx.x; t < O * C; t += 256) Ws[t] = a.W[t];
    const int d = threadIdx.x % Dp, hg = threadIdx.x / Dp;
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.F0 * D; t += 256) x0[t] = a.X0[b * a.x0_ld + t];
        for (int t = threadIdx.x; t < a.Mi * D; t += 256) xi[t] = a.Xi[b * a.xi_ld + t];
        for (int t = threadIdx.x; t < O * D; t += 256) {
            float v = a.dXn ? a.dXn[b * O * D + t] : 0.f;
            if (a.dpool) v += a.dpool[b * a.dpool_ld + t / D];
            g[t] = v;
        }
        for (int t = threadIdx.x; t < NH * a.Mi * D; t += 256) red[t] = 0.f;
        __syncthreads();
        if (d < D) {
            for (int h = hg; h < a.F0; h += NH) {
                const float xh = x0[h * D + d];
                float dx0 = 0.f;
                for (int m = 0; m < a.Mi; ++m) {
                    float t = 0.f;
                    for (int o = 0; o < O; ++o) t = fmaf(g[o * D + d], Ws[o * C + h * a.Mi + m], t);
                    dx0 = fmaf(t, xi[m * D + d], dx0);
                    red[(hg * a.Mi + m) * D + d] = fmaf(t, xh, red[(hg * a.Mi + m) * D + d]);
                }
                float* o0 = a.dX0 + b * a.dx0_ld + h * D + d;
                *o0 = a.acc_dx0 ? *o0 + dx0 : dx0;
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < a.Mi * D; t += 256) {
            float s = 0.f;
            for (int k = 0; k < NH; ++k) s += red[k * a.Mi * D + t];
            a.dXi[b * a.dxi_ld + t] = s;
        }
    }
}

// ---- backward: weight gradient partials ----------------------------------------------------------
// workgroup G accumulates dW[o,c] (and dbias[o]) of its samples in LDS; entry e is owned by thread
// e % 256 (no atomics).  partial[G][O*C + O]; a column sum over G finishes the reduction.
__global__ __launch_bounds__(256) void k_cin_bwd_dw(CinArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C = a.F0 * a.Mi, D = a.D, O = a.O;
    float* dW = smem;                    // [O*C + O]
    float* x0 = dW + O * C + O;          // [F0*D]
    float* xi = x0 + a.F0 * D;           // [Mi*D]
    float* g = xi + a.Mi * D;            // [O*D]
    for (int t = threadIdx.x; t < O * C + O; t += 256) dW[t] = 0.f;
    for (int64_t b = blockIdx.x; b < a.B; b += gridDim.x) {
        __syncthreads();
        for (int t = threadIdx.x; t < a.F0 * D; t += 256) x0[t] = a.X0[b * a.x0_ld + t];
        for (int t = threadIdx.x; t < a.Mi * D; t += 256) xi[t] = a.Xi[b * a.xi_ld + t];
        for (int t = threadIdx.x; t < O * D; t += 256) {
            float v = a.dXn ? a.dXn[b * O * D + t] : 0.f;
            if (a.dpool) v += a.dpool[b * a.dpool_ld + t / D];
            g[t] = v;
        }
        __syncthreads();
        for (int e = threadIdx.x; e < O * C; e += 256) {
            const int o = e / C, c = e - o * C;
            const int h = c / a.Mi, m = c - h * a.Mi;
            float s = 0.f;
            for (int d = 0; d < D; ++d) s = fmaf(g[o * D + d] * x0[h * D + d], xi[m * D + d], s);
            dW[e] += s;
        }
        for (int o = threadIdx.x; o < O; o += 256) {
            float s = 0.f;
            for (int d = 0; d < D; ++d) s += g[o * D + d];
            dW[O * C + o] += s;
        }
    }
    __syncthreads();
    float* out = a.partial + (int64_t)blockIdx.x * (O * C + O);
    for (int t = threadIdx.x; t < O * C + O; t += 256) out[t] = dW[t];
}

static int fx_cin_check(const char* who, int F0, int Mi, int D, int O) {
    FX_CHECK_ARG(F0 >= 1 && Mi >= 1 && D >= 1 && O >= 1, "%s: bad sizes", who);
    FX_CHECK_ARG((int64_t)O * F0 * Mi + O <= FX_CIN_MAX_W_FLOATS,
                 "%s: O*F0*Mi = %lld floats exceed the LDS-resident limit (%d); split O", who,
                 (long long)O * F0 * Mi, FX_CIN_MAX_W_FLOATS);
    FX_CHECK_ARG(F0 * D <= FX_CIN_MAX_TILE && Mi * D <= FX_CIN_MAX_TILE && O * D <= FX_CIN_MAX_TILE &&
                     D <= 256,
                 "%s: tile too large (F0=%d Mi=%d D=%d O=%d)", who, F0, Mi, D, O);
    return FX_OK;
}

extern "C" int64_t fx_cin_workgroups(void) { return 256; }

extern "C" int fx_cin_fwd(const float* X0, int64_t x0_ld, int32_t F0, const float* Xi,
                          int64_t xi_ld, int32_t Mi, int32_t D, const float* W, const float* bias,
                          int32_t O, float* Xn, float* pool, int64_t pool_ld, int64_t B,
                          fx_stream_t stream) {
    if (fx_cin_check("fx_cin_fwd", F0, Mi, D, O) != FX_OK) return FX_ERR_INVALID;
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(X0 && Xi && W && bias && Xn, "fx_cin_fwd: null pointer");
    CinArgs a;
    memset(&a, 0, sizeof(a));
    a.X0 = X0; a.x0_ld = x0_ld; a.Xi = Xi; a.xi_ld = xi_ld; a.W = W; a.bias = bias; a.Xn = Xn;
    a.pool = pool; a.pool_ld = pool_ld; a.B = B; a.F0 = F0; a.Mi = Mi; a.D = D; a.O = O;
    const size_t lds = sizeof(float) * ((size_t)O * F0 * Mi + (size_t)(F0 + Mi) * D);
    FX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cin_fwd),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int64_t grid = B < 256 ? B : 256;
    hipLaunchKernelGGL(k_cin_fwd, dim3((unsigned)grid), dim3(256), lds, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_cin_bwd(const float* X0, int64_t x0_ld, int32_t F0, const float* Xi,
                          int64_t xi_ld, int32_t Mi, int32_t D, const float* W, int32_t O,
                          const float* dXn, const float* dpool, int64_t dpool_ld, float* dX0,
                          int64_t dx0_ld, int32_t accumulate_dx0, float* dXi, int64_t dxi_ld,
                          float* partial, int64_t B, fx_stream_t stream) {
    if (fx_cin_check("fx_cin_bwd", F0, Mi, D, O) != FX_OK) return FX_ERR_INVALID;
    FX_CHECK_ARG(B >= 1, "fx_cin_bwd: B must be >= 1");
    FX_CHECK_ARG(X0 && Xi && W && (dXn || dpool) && dX0 && dXi && partial,
                 "fx_cin_bwd: null pointer");
    CinArgs a;
    memset(&a, 0, sizeof(a));
    a.X0 = X0; a.x0_ld = x0_ld; a.Xi = Xi; a.xi_ld = xi_ld; a.W = W; a.dXn = dXn; a.dpool = dpool;
    a.dpool_ld = dpool_ld; a.dX0 = dX0; a.dx0_ld = dx0_ld; a.acc_dx0 = accumulate_dx0; a.dXi = dXi;
    a.dxi_ld = dxi_ld; a.partial = partial; a.B = B; a.F0 = F0; a.Mi = Mi; a.D = D; a.O = O;
    int Dp = 1;
    while (Dp < D) Dp <<= 1;
    const int NH = 256 / Dp;
    const size_t lds_dx = sizeof(float) * ((size_t)O * F0 * Mi + (size_t)(F0 + Mi + O) * D +
                                           (size_t)NH * Mi * D);
    const size_t lds_dw = sizeof(float) * ((size_t)O * F0 * Mi + O + (size_t)(F0 + Mi + O) * D);
    FX_CHECK_ARG(lds_dx <= 160 * 1024 && lds_dw <= 160 * 1024, "fx_cin_bwd: LDS need too large");
    FX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cin_bwd_dx),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dx));
    FX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cin_bwd_dw),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw));
    hipStream_t s = fx_hip_stream(stream);
    const int64_t grid = B < 256 ? B : 256;
    hipLaunchKernelGGL(k_cin_bwd_dx, dim3((unsigned)grid), dim3(256), lds_dx, s, a);
    // weight-gradient partials always use the full 256 workgroups so `partial` has a fixed shape
    hipLaunchKernelGGL(k_cin_bwd_dw, dim3(256), dim3(256), lds_dw, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
