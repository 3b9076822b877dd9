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

#include <stdlib.h>

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
    const float* wimg;       // fx_cin_pack_w's LDS images of W (MFMA kernels), or null
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

// =================================================================================================
// Second-generation kernels (used whenever Mi <= 64, and O <= 32 for the input gradient).
// The first-generation kernels above run one 4-wave workgroup per CU with two or three dependent LDS
// reads per FMA and no unrolling: latency-bound at 2 % of the VALU rate (layer 1 at B = 4096:
// 0.9 / 1.2 / 1.8 ms for fwd / dX / dW).  Here: 16 waves per CU (1024 threads = 4 sample slots),
// W rows padded to a multiple of 4 so the m direction is read with ds_read_b128, the per-thread
// operands that do not change in the inner loop (a column of Xi, a column of g, the dW / dXi
// accumulators) live in registers, loops over m are fully unrolled (template MI).
// =================================================================================================
// Static LDS (the whole 160 KB a workgroup may have): a launch with > 64 KB of DYNAMIC LDS was
// measured to cost ~13 us more per launch inside the step graph (profiles/r01_gemm_pipe.txt).
#define FX_CIN2_LDS_FLOATS 40000

template <int MI>
__global__ __launch_bounds__(1024) void k_cin_fwd2(CinArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[FX_CIN2_LDS_FLOATS];
    const int D = a.D, O = a.O, F0 = a.F0, Mi = a.Mi;
    const int MiP = (Mi + 3) & ~3;
    float* Ws = smem;                                        // [O][F0][MiP]
    const int slot = threadIdx.x >> 8, t = threadIdx.x & 255;
    float* x0 = Ws + O * F0 * MiP + slot * ((F0 + Mi + O) * D);   // [F0*D]
    float* xi = x0 + F0 * D;                                 // [Mi*D]
    float* xo = xi + Mi * D;                                 // [O*D]  (this sample's Xn, for the pool)
    for (int e = threadIdx.x; e < O * F0 * MiP; e += 1024) {
        const int m = e % MiP, oh = e / MiP;
        Ws[e] = m < Mi ? a.W[(int64_t)oh * Mi + m] : 0.f;
    }
    const int64_t per_round = (int64_t)gridDim.x * 4;
    const int64_t rounds = (a.B + per_round - 1) / per_round;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t b = (it * gridDim.x + blockIdx.x) * 4 + slot;
        const bool valid = b < a.B;
        __syncthreads();
        if (valid) {
            for (int e = t; e < F0 * D; e += 256) x0[e] = a.X0[b * a.x0_ld + e];
            for (int e = t; e < Mi * D; e += 256) xi[e] = a.Xi[b * a.xi_ld + e];
        }
        __syncthreads();
        if (valid) {
            for (int od = t; od < O * D; od += 256) {
                const int o = od / D, d = od - o * D;
                float xr[MI];
#pragma unroll
                for (int m = 0; m < MI; ++m) xr[m] = m < Mi ? xi[m * D + d] : 0.f;
                const float* w = Ws + o * F0 * MiP;
                float acc = 0.f;
                for (int h = 0; h < F0; ++h) {
                    float sacc = 0.f;
#pragma unroll
                    for (int m4 = 0; m4 < MI; m4 += 4) {
                        if (m4 < MiP) {
                            const float4 wq = *reinterpret_cast<const float4*>(w + h * MiP + m4);
                            sacc = fmaf(wq.x, xr[m4], sacc);
                            if (m4 + 1 < MI) sacc = fmaf(wq.y, xr[m4 + 1], sacc);
                            if (m4 + 2 < MI) sacc = fmaf(wq.z, xr[m4 + 2], sacc);
                            if (m4 + 3 < MI) sacc = fmaf(wq.w, xr[m4 + 3], sacc);
                        }
                    }
                    acc = fmaf(x0[h * D + d], sacc, acc);
                }
                acc += a.bias[o];
                a.Xn[(b * O + o) * D + d] = acc;
                xo[od] = acc;
            }
        }
        if (a.pool) {
            __syncthreads();
            if (valid) {
                for (int o = t; o < O; o += 256) {
                    float sum = 0.f;
                    for (int d = 0; d < D; ++d) sum += xo[o * D + d];
                    a.pool[b * a.pool_ld + o] = sum;
                }
            }
        }
    }
}

// input gradients.  thread = (slot, hg, d): g[:,d] and Xi[:,d] in registers, T[h, m..m+3] from
// ds_read_b128 rows of W, dXi accumulated in registers and combined over the h-groups by wave
// shuffles + one LDS pass over the 4 waves of the slot.
template <int MI, int OM>
__global__ __launch_bounds__(1024) void k_cin_bwd_dx2(CinArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[FX_CIN2_LDS_FLOATS];
    const int D = a.D, O = a.O, F0 = a.F0, Mi = a.Mi;
    const int MiP = (Mi + 3) & ~3;
    int Dp = 1;
    while (Dp < D) Dp <<= 1;
    const int NH = 256 / Dp;
    float* Ws = smem;                                        // [O][F0][MiP]
    const int slot = threadIdx.x >> 8, t = threadIdx.x & 255;
    float* red = Ws + O * F0 * MiP + slot * (4 * Mi * D);    // [4 waves][Mi*D]
    for (int e = threadIdx.x; e < O * F0 * MiP; e += 1024) {
        const int m = e % MiP, oh = e / MiP;
        Ws[e] = m < Mi ? a.W[(int64_t)oh * Mi + m] : 0.f;
    }
    const int d = t % Dp, hg = t / Dp;
    const int wave = t >> 6;
    const int64_t per_round = (int64_t)gridDim.x * 4;
    const int64_t rounds = (a.B + per_round - 1) / per_round;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t b = (it * gridDim.x + blockIdx.x) * 4 + slot;
        const bool valid = b < a.B;
        const bool on = valid && d < D;
        __syncthreads();                                     // W staged / previous `red` consumed
        float gr[OM], xr[MI], dxi[MI];
#pragma unroll
        for (int o = 0; o < OM; ++o) {
            float v = 0.f;
            if (on && o < O) {
                if (a.dXn) v = a.dXn[(b * O + o) * D + d];
                if (a.dpool) v += a.dpool[b * a.dpool_ld + o];
            }
            gr[o] = v;
        }
#pragma unroll
        for (int m = 0; m < MI; ++m) {
            xr[m] = (on && m < Mi) ? a.Xi[b * a.xi_ld + m * D + d] : 0.f;
            dxi[m] = 0.f;
        }
        if (on) {
            for (int h = hg; h < F0; h += NH) {
                const float xh = a.X0[b * a.x0_ld + h * D + d];
                float dx0 = 0.f;
#pragma unroll
                for (int m4 = 0; m4 < MI; m4 += 4) {
                    if (m4 < MiP) {
                        float4 tq = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                        for (int o = 0; o < OM; ++o) {
                            if (o < O) {
                                const float4 wq = *reinterpret_cast<const float4*>(
                                    Ws + (o * F0 + h) * MiP + m4);
                                tq.x = fmaf(gr[o], wq.x, tq.x);
                                tq.y = fmaf(gr[o], wq.y, tq.y);
                                tq.z = fmaf(gr[o], wq.z, tq.z);
                                tq.w = fmaf(gr[o], wq.w, tq.w);
                            }
                        }
                        dx0 = fmaf(tq.x, xr[m4], dx0);
                        dxi[m4] = fmaf(tq.x, xh, dxi[m4]);
                        if (m4 + 1 < MI) { dx0 = fmaf(tq.y, xr[m4 + 1], dx0); dxi[m4 + 1] = fmaf(tq.y, xh, dxi[m4 + 1]); }
                        if (m4 + 2 < MI) { dx0 = fmaf(tq.z, xr[m4 + 2], dx0); dxi[m4 + 2] = fmaf(tq.z, xh, dxi[m4 + 2]); }
                        if (m4 + 3 < MI) { dx0 = fmaf(tq.w, xr[m4 + 3], dx0); dxi[m4 + 3] = fmaf(tq.w, xh, dxi[m4 + 3]); }
                    }
                }
                float* o0 = a.dX0 + b * a.dx0_ld + h * D + d;
                *o0 = a.acc_dx0 ? *o0 + dx0 : dx0;
            }
        }
        // h-groups inside a wave (lanes d + Dp*k): xor shuffles, fixed order
#pragma unroll
        for (int m = 0; m < MI; ++m) {
            float v = dxi[m];
            for (int off = Dp; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            dxi[m] = v;
        }
        if ((t & 63) < Dp && d < D) {
#pragma unroll
            for (int m = 0; m < MI; ++m)
                if (m < Mi) red[(wave * Mi + m) * D + d] = dxi[m];
        }
        __syncthreads();
        if (valid) {
            for (int e = t; e < Mi * D; e += 256) {       // the slot's 4 waves, fixed order
                float sum = 0.f;
                for (int w = 0; w < 4; ++w) sum += red[w * Mi * D + e];
                a.dXi[b * a.dxi_ld + e] = sum;
            }
        }
    }
}

// weight-gradient partials.  thread = (o, h) pair, dW[o,h,0..Mi) in registers over all samples of
// the workgroup; Xi is staged transposed ([d][MiP]) so the m direction is a ds_read_b128 broadcast.
template <int MI>
__global__ __launch_bounds__(1024) void k_cin_bwd_dw2(CinArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[FX_CIN2_LDS_FLOATS];
    const int D = a.D, O = a.O, F0 = a.F0, Mi = a.Mi, C = F0 * Mi;
    const int MiP = (Mi + 3) & ~3;
    constexpr int SB = 4;                                    // samples staged per round
    const int per = (D * MiP + (F0 + O) * (D + 1) + 3) & ~3;  // floats per staged sample (16-B multiple)
    float* stage = smem;
    const int pair = threadIdx.x;
    const bool own = pair < O * F0;
    const int o = own ? pair / F0 : 0, h = own ? pair % F0 : 0;
    float acc[MI];
#pragma unroll
    for (int m = 0; m < MI; ++m) acc[m] = 0.f;
    float bacc = 0.f;
    const int64_t per_round = (int64_t)gridDim.x * SB;
    const int64_t rounds = (a.B + per_round - 1) / per_round;
    for (int64_t it = 0; it < rounds; ++it) {
        const int64_t b0 = (it * gridDim.x + blockIdx.x) * SB;
        __syncthreads();
        for (int sb = 0; sb < SB; ++sb) {
            const int64_t b = b0 + sb;
            if (b >= a.B) break;
            float* xt = stage + sb * per;                    // [D][MiP]   xt[d*MiP + m] = Xi[m,d]
            float* x0 = xt + D * MiP;                        // [F0][D+1]
            float* g = x0 + F0 * (D + 1);                    // [O][D+1]
            for (int e = threadIdx.x; e < Mi * D; e += 1024) {
                const int m = e / D, d = e - m * D;
                xt[d * MiP + m] = a.Xi[b * a.xi_ld + e];
            }
            for (int e = threadIdx.x; e < D * (MiP - Mi); e += 1024) {
                const int d = e / (MiP - Mi), m = Mi + e % (MiP - Mi);
                xt[d * MiP + m] = 0.f;
            }
            for (int e = threadIdx.x; e < F0 * D; e += 1024)
                x0[(e / D) * (D + 1) + e % D] = a.X0[b * a.x0_ld + e];
            for (int e = threadIdx.x; e < O * D; e += 1024) {
                float v = a.dXn ? a.dXn[b * O * D + e] : 0.f;
                if (a.dpool) v += a.dpool[b * a.dpool_ld + e / D];
                g[(e / D) * (D + 1) + e % D] = v;
            }
        }
        __syncthreads();
        if (own) {
            for (int sb = 0; sb < SB; ++sb) {
                if (b0 + sb >= a.B) break;
                const float* xt = stage + sb * per;
                const float* x0 = xt + D * MiP;
                const float* g = x0 + F0 * (D + 1);
                for (int d = 0; d < D; ++d) {
                    const float gd = g[o * (D + 1) + d];
                    const float u = gd * x0[h * (D + 1) + d];
                    if (h == 0) bacc += gd;
#pragma unroll
                    for (int m4 = 0; m4 < MI; m4 += 4) {
                        if (m4 < MiP) {
                            const float4 xq = *reinterpret_cast<const float4*>(xt + d * MiP + m4);
                            acc[m4] = fmaf(u, xq.x, acc[m4]);
                            if (m4 + 1 < MI) acc[m4 + 1] = fmaf(u, xq.y, acc[m4 + 1]);
                            if (m4 + 2 < MI) acc[m4 + 2] = fmaf(u, xq.z, acc[m4 + 2]);
                            if (m4 + 3 < MI) acc[m4 + 3] = fmaf(u, xq.w, acc[m4 + 3]);
                        }
                    }
                }
            }
        }
    }
    float* out = a.partial + (int64_t)blockIdx.x * (O * C + O);
    if (own) {
#pragma unroll
        for (int m = 0; m < MI; ++m)
            if (m < Mi) out[(o * F0 + h) * Mi + m] = acc[m];
        if (h == 0) out[O * C + o] = bacc;
    }
}

// =================================================================================================
// Round 3: the CIN products on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulate, the same 64 flop/clk/SIMD peak as the 32x32x2 form) for the shape class of the BASELINE
// configuration: D = 16 (the 16 dims of ONE sample are the 16 columns of an MFMA tile), O <= 16 (rows),
// Mi <= 48.  The compress GEMM of compressed_interaction_net.py:70-74,
//     Xn[o, d] = sum_{h,m} W[o, (h,m)] * (X0[h,d] * Xi[m,d]),
// is W [16 x F0*Mi] times Z [F0*Mi x 16] per sample; Z is never stored: the outer-product element a
// lane needs for its B fragment is ONE multiply of two registers (x0[h][d] * xi[m][d]).  A wave carries
// NS samples through the K loop, so a W fragment read from LDS feeds NS MFMAs.
//   lane l: r = l & 15 (row of A / column of B, D), kk = l >> 4 (k of A and B; D holds rows 4*kk + i)
//   forward   A = W[o = r][k = (h, 4*mq + kk)]            B = x0[h][d = r] * xi[4*mq + kk][d = r]
//   backward  T[(h,m), d] = sum_o W[o,(h,m)] g[o,d]:      A = W^T[m = 16*mt + r][o = 4*j + kk], B = g[4*j + kk][r]
//             dX0[h,d] = sum_m T xi[m,d] (registers + two shuffles), dXi[m,d] = sum_h T x0[h,d] (registers)
//   weights   dW[o,(h,m)] = sum_{b,d} g[b,o,d] Z[b,(h,m),d]: A = g[o = r][d = 4*j + kk],
//             B = x0[h][4*j + kk] * xi[16*mt + r][4*j + kk]; a workgroup's four waves split the k tiles and
//             keep their dW tiles in accumulators over all its samples -> partial[G] as before.
// fp32 VALU kernels above stay for every other shape.
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FX_CIN_MFMA_NS 4

static int fx_cin_mfma_mode() {     // FX_CIN_MFMA=0: the VALU kernels everywhere (A/B runs)
    static const int mode = []() {
        const char* e = getenv("FX_CIN_MFMA");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

#define FX_CIN_HB 5      // rows of X0 fetched per register block (two blocks in flight)

// the two instantiations: Mi <= 16 (layers fed by a 16-map layer) and Mi <= 40 (layer 1 of 39/40 fields)
__host__ __device__ __forceinline__ int fx_cin_mq(int Mi) { return Mi <= 16 ? 4 : 10; }
__host__ __device__ __forceinline__ int fx_cin_mt(int Mi) { return Mi <= 16 ? 1 : 3; }

// The LDS images of W, laid out so that a wave's A fragment is one conflict-free ds_read_b32:
//   forward  [q = h*MQ + mq][lane]        = W[o = r][h*Mi + 4*mq + kk]          MQ = ceil(Mi / 4)
//   dX       [tile = h*MT + mt][j][lane]  = W[o = 4*j + kk][h*Mi + 16*mt + r]   MT = ceil(Mi / 16)
// zero where o >= O or m >= Mi.  fx_cin_pack_w writes both once per step into w_img (forward image
// first), and every workgroup copies its image with float4 loads; without w_img a workgroup gathers
// the image from W itself (same values, ~100 dependent-latency loads per thread: the slow start).
__device__ __forceinline__ float fx_cin_wq_elem(const float* W, int F0, int Mi, int O, int MQ, int e) {
    const int q = e >> 6, l = e & 63;
    const int o = l & 15, m = (q % MQ) * 4 + (l >> 4), h = q / MQ;
    return (o < O && m < Mi) ? W[(int64_t)o * F0 * Mi + h * Mi + m] : 0.f;
}

__device__ __forceinline__ float fx_cin_wb_elem(const float* W, int F0, int Mi, int O, int MT, int e) {
    const int tile = e >> 8, j = (e >> 6) & 3, l = e & 63;
    const int o = 4 * j + (l >> 4), m = (tile % MT) * 16 + (l & 15), h = tile / MT;
    return (o < O && m < Mi) ? W[(int64_t)o * F0 * Mi + h * Mi + m] : 0.f;
}

__global__ __launch_bounds__(256) void k_cin_pack_w(const float* W, int F0, int Mi, int O, int MQ, int MT,
                                                    float* img) {
    const int nf = F0 * MQ * 64, nd = F0 * MT * 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nf + nd; e += gridDim.x * 256)
        img[e] = e < nf ? fx_cin_wq_elem(W, F0, Mi, O, MQ, e) : fx_cin_wb_elem(W, F0, Mi, O, MT, e - nf);
}

__device__ __forceinline__ void fx_cin_copy_img(float* lds, const float* img, int n_floats) {
    const float4* src = reinterpret_cast<const float4*>(img);
    float4* dst = reinterpret_cast<float4*>(lds);
#pragma unroll 8
    for (int e = threadIdx.x; e < (n_floats >> 2); e += 256) dst[e] = src[e];
}

// MQ = quads of m per h (Mi padded to 4*MQ); WF = floats of the LDS image of W (F0 * MQ * 64)
template <int MQ, int WF>
__global__ __launch_bounds__(256) void k_cin_fwd_mfma(CinArgs a) {
    constexpr int NS = FX_CIN_MFMA_NS, HB = FX_CIN_HB;
    __shared__ __attribute__((aligned(16))) float Wq[WF];
    const int F0 = a.F0, Mi = a.Mi, O = a.O;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kk = lane >> 4;
    const int nq = F0 * MQ;
    if (a.wimg) {
        fx_cin_copy_img(Wq, a.wimg, nq * 64);
    } else {
        for (int e = threadIdx.x; e < nq * 64; e += 256) Wq[e] = fx_cin_wq_elem(a.W, F0, Mi, O, MQ, e);
    }
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = (4 * kk + i < O) ? a.bias[4 * kk + i] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * 4 * NS;
    bool staged = false;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * NS; base < a.B; base += stride) {
        const float* x0p[NS];
        float xi[NS][MQ];
        f32x4 acc[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s < a.B ? base + s : a.B - 1;      // clamped: stores are guarded
            x0p[s] = a.X0 + b * a.x0_ld + r;
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq) {
                const int m = 4 * mq + kk;
                xi[s][mq] = m < Mi ? a.Xi[b * a.xi_ld + m * 16 + r] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[s][i] = 0.f;
        }
        float xa[NS][HB], xb[NS][HB];
        auto loadblk = [&](float (&x)[NS][HB], int h0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) {
                    const int h = h0 + hh < F0 ? h0 + hh : F0 - 1;
                    x[s][hh] = x0p[s][h * 16];
                }
        };
        auto compute = [&](const float (&x)[NS][HB], int h0) {
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                if (h0 + hh < F0) {
                    const float* wq = Wq + ((h0 + hh) * MQ) * 64 + lane;
#pragma unroll
                    for (int mq = 0; mq < MQ; ++mq) {
                        const float wa = wq[mq * 64];
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, x[s][hh] * xi[s][mq], acc[s],
                                                                          0, 0, 0);
                    }
                }
            }
        };
        loadblk(xa, 0);
        if (!staged) {                   // the operand loads above are in flight behind the W image
            __syncthreads();
            staged = true;
        }
        for (int h0 = 0; h0 < F0; h0 += 2 * HB) {
            loadblk(xb, h0 + HB);
            compute(xa, h0);
            loadblk(xa, h0 + 2 * HB);
            compute(xb, h0 + HB);
        }
        // D: rows o = 4*kk + i, column d = r
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 4 * kk + i;
                const float v = acc[s][i] + bias4[i];
                float sum = v;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                if (b < a.B && o < O) {
                    a.Xn[(b * O + o) * 16 + r] = v;
                    if (a.pool && r == 0) a.pool[b * a.pool_ld + o] = sum;
                }
            }
        }
    }
    if (!staged) __syncthreads();
}

// MT = 16-wide m tiles per h (Mi padded to 16*MT); WF = floats of the LDS image (F0 * MT * 256)
template <int MT, int WF>
__global__ __launch_bounds__(256) void k_cin_dx_mfma(CinArgs a) {
    constexpr int NS = FX_CIN_MFMA_NS, HB = FX_CIN_HB;
    __shared__ __attribute__((aligned(16))) float Wb[WF];
    const int F0 = a.F0, Mi = a.Mi, O = a.O;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kk = lane >> 4;
    const int nt = F0 * MT;
    if (a.wimg) {
        fx_cin_copy_img(Wb, a.wimg + (int64_t)F0 * fx_cin_mq(Mi) * 64, nt * 256);
    } else {
        for (int e = threadIdx.x; e < nt * 256; e += 256) Wb[e] = fx_cin_wb_elem(a.W, F0, Mi, O, MT, e);
    }
    const bool accd = a.acc_dx0 != 0;
    const int64_t stride = (int64_t)gridDim.x * 4 * NS;
    bool staged = false;
    for (int64_t base = ((int64_t)blockIdx.x * 4 + wave) * NS; base < a.B; base += stride) {
        const float* x0p[NS];
        float* d0p[NS];
        float g[NS][4], xi[NS][MT][4], dxi[NS][MT][4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s < a.B ? base + s : a.B - 1;
            x0p[s] = a.X0 + b * a.x0_ld + r;
            d0p[s] = a.dX0 + b * a.dx0_ld + r;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 4 * j + kk;
                float v = 0.f;
                if (o < O) {
                    if (a.dXn) v = a.dXn[(b * O + o) * 16 + r];
                    if (a.dpool) v += a.dpool[b * a.dpool_ld + o];
                }
                g[s][j] = v;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * mt + 4 * kk + i;
                    xi[s][mt][i] = m < Mi ? a.Xi[b * a.xi_ld + m * 16 + r] : 0.f;
                    dxi[s][mt][i] = 0.f;
                }
        }
        // X0 rows (and, when dX0 accumulates, the values already there) travel in register blocks of HB
        float xa[NS][HB], xb[NS][HB], oa[NS][HB], ob[NS][HB];
        auto loadblk = [&](float (&x)[NS][HB], float (&old)[NS][HB], int h0) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) {
                    const int h = h0 + hh < F0 ? h0 + hh : F0 - 1;
                    x[s][hh] = x0p[s][h * 16];
                    old[s][hh] = accd ? d0p[s][h * 16] : 0.f;
                }
        };
        auto compute = [&](const float (&x)[NS][HB], const float (&old)[NS][HB], int h0) {
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                const int h = h0 + hh;
                if (h < F0) {
                    float dx0[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s) dx0[s] = 0.f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float* wb = Wb + (h * MT + mt) * 256 + lane;
                        f32x4 T[NS];
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int i = 0; i < 4; ++i) T[s][i] = 0.f;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float wa = wb[j * 64];
#pragma unroll
                            for (int s = 0; s < NS; ++s)
                                T[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, g[s][j], T[s], 0, 0, 0);
                        }
                        // T[s][i] = T[(h, m = 16*mt + 4*kk + i), d = r]
#pragma unroll
                        for (int s = 0; s < NS; ++s)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                dx0[s] = fmaf(T[s][i], xi[s][mt][i], dx0[s]);
                                dxi[s][mt][i] = fmaf(T[s][i], x[s][hh], dxi[s][mt][i]);
                            }
                    }
#pragma unroll
                    for (int s = 0; s < NS; ++s) {
                        float v = dx0[s];
                        v += __shfl_xor(v, 16, 64);
                        v += __shfl_xor(v, 32, 64);
                        if (base + s < a.B && kk == 0) d0p[s][h * 16] = old[s][hh] + v;
                    }
                }
            }
        };
        loadblk(xa, oa, 0);
        if (!staged) {
            __syncthreads();
            staged = true;
        }
        for (int h0 = 0; h0 < F0; h0 += 2 * HB) {
            loadblk(xb, ob, h0 + HB);
            compute(xa, oa, h0);
            loadblk(xa, oa, h0 + 2 * HB);
            compute(xb, ob, h0 + HB);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s;
            if (b >= a.B) continue;
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * mt + 4 * kk + i;
                    if (m < Mi) a.dXi[b * a.dxi_ld + m * 16 + r] = dxi[s][mt][i];
                }
        }
    }
    if (!staged) __syncthreads();
}

#define FX_CIN_DW_SB 4     // samples staged per barrier in the dW kernel

// TPW = k tiles per wave = MT * ceil(F0 / 4): wave w owns the rows h = w, w + 4, ... of X0
template <int MT, int TPW>
__global__ __launch_bounds__(256) void k_cin_dw_mfma(CinArgs a) {
    constexpr int SB = FX_CIN_DW_SB;
    __shared__ float x0s[2][SB][48 * 16];  // a sample's X0 [F0 <= 40][16], double-buffered
    __shared__ float gs[2][SB][16 * 16];   // g[o][d]
    __shared__ float xis[2][SB][48 * 17];  // Xi rows padded to 17 floats (conflict-free column reads)
    const int F0 = a.F0, Mi = a.Mi, O = a.O, C = F0 * Mi;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kk = lane >> 4;
    f32x4 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    float db = 0.f;                        // wave 0: sum over (b, d = 4*j + kk) of g[o = r][d]
    // SB samples' operands travel global -> registers (issued before the MFMA section) -> LDS (after
    // it); sample sb of round k is b = (k * SB + sb) * gridDim.x + blockIdx.x, absent ones are zeros
    float px0[SB][3], pxi[SB][MT], pg[SB];
    auto fetch = [&](int64_t k) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
            const int64_t b = (k * SB + sb) * gridDim.x + blockIdx.x;
            const bool live = b < a.B;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int e = threadIdx.x + 256 * q;
                px0[sb][q] = (live && e < F0 * 16) ? a.X0[b * a.x0_ld + e] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < MT; ++q) {
                const int m = (threadIdx.x >> 4) + 16 * q;
                pxi[sb][q] = (live && m < Mi) ? a.Xi[b * a.xi_ld + m * 16 + (threadIdx.x & 15)] : 0.f;
            }
            const int o = threadIdx.x >> 4;
            float v = 0.f;
            if (live && o < O) {
                if (a.dXn) v = a.dXn[b * O * 16 + threadIdx.x];
                if (a.dpool) v += a.dpool[b * a.dpool_ld + o];
            }
            pg[sb] = v;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
#pragma unroll
            for (int q = 0; q < 3; ++q) x0s[buf][sb][threadIdx.x + 256 * q] = px0[sb][q];
#pragma unroll
            for (int q = 0; q < MT; ++q)
                xis[buf][sb][((threadIdx.x >> 4) + 16 * q) * 17 + (threadIdx.x & 15)] = pxi[sb][q];
            gs[buf][sb][threadIdx.x] = pg[sb];
        }
    };
    const int64_t per_round = (int64_t)SB * gridDim.x;
    const int64_t rounds = (a.B - blockIdx.x + per_round - 1) / per_round;   // blockIdx.x < B or 0 rounds
    int buf = 0;
    if (rounds > 0) {
        fetch(0);
        put(0);
    }
    __syncthreads();
    for (int64_t k = 0; k < rounds; ++k) {
        if (k + 1 < rounds) fetch(k + 1);
#pragma unroll 1
        for (int sb = 0; sb < SB; ++sb) {
            float ga[4], xr[MT][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ga[j] = gs[buf][sb][r * 16 + 4 * j + kk];
                if (wave == 0) db += ga[j];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[mt][j] = xis[buf][sb][(16 * mt + r) * 17 + 4 * j + kk];
#pragma unroll
            for (int t = 0; t < TPW; ++t) {    // tile t of this wave: h = wave + 4 * (t / MT), mt = t % MT
                const int h = wave + 4 * (t / MT);
                if (h < F0) {
                    const float* xh = x0s[buf][sb] + h * 16 + kk;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(ga[j], xh[4 * j] * xr[t % MT][j],
                                                                      acc[t], 0, 0, 0);
                }
            }
        }
        if (k + 1 < rounds) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // partial[blockIdx][o * C + h * Mi + m]: D rows o = 4*kk + i, column = m offset r
    float* part = a.partial + (int64_t)blockIdx.x * ((int64_t)O * C + O);
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int h = wave + 4 * (t / MT);
        if (h < F0) {
            const int m = (t % MT) * 16 + r;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 4 * kk + i;
                if (o < O && m < Mi) part[(int64_t)o * C + h * Mi + m] = acc[t][i];
            }
        }
    }
    if (wave == 0) {
        db += __shfl_xor(db, 16, 64);
        db += __shfl_xor(db, 32, 64);
        if (kk == 0 && r < O) part[(int64_t)O * C + r] = db;
    }
}

// the MFMA class: D = 16, O <= 16, F0 <= 40, Mi <= 40
static bool fx_cin_mfma_ok(int32_t F0, int32_t Mi, int32_t D, int32_t O) {
    return fx_cin_mfma_mode() && D == 16 && O >= 1 && O <= 16 && F0 >= 1 && F0 <= 40 &&
           Mi >= 1 && Mi <= 40;
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

extern "C" int64_t fx_cin_wimg_floats(int32_t F0, int32_t Mi, int32_t D, int32_t O) {
    if (!fx_cin_mfma_ok(F0, Mi, D, O)) return 0;
    return (int64_t)F0 * fx_cin_mq(Mi) * 64 + (int64_t)F0 * fx_cin_mt(Mi) * 256;
}

extern "C" int fx_cin_pack_w(const float* W, int32_t F0, int32_t Mi, int32_t D, int32_t O, float* w_img,
                             fx_stream_t stream) {
    FX_CHECK_ARG(W && w_img, "fx_cin_pack_w: null pointer");
    FX_CHECK_ARG(((uintptr_t)w_img & 15) == 0, "fx_cin_pack_w: w_img must be 16-byte aligned");
    const int64_t n = fx_cin_wimg_floats(F0, Mi, D, O);
    FX_CHECK_ARG(n > 0, "fx_cin_pack_w: no image for this shape (fx_cin_wimg_floats == 0)");
    hipLaunchKernelGGL(k_cin_pack_w, dim3((unsigned)fx_ceil_div(n, 256)), dim3(256), 0, fx_hip_stream(stream),
                       W, F0, Mi, O, fx_cin_mq(Mi), fx_cin_mt(Mi), w_img);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_cin_fwd(const float* X0, int64_t x0_ld, int32_t F0, const float* Xi,
                          int64_t xi_ld, int32_t Mi, int32_t D, const float* W, const float* bias,
                          int32_t O, float* Xn, float* pool, int64_t pool_ld, int64_t B,
                          const float* w_img, fx_stream_t stream) {
    if (fx_cin_check("fx_cin_fwd", F0, Mi, D, O) != FX_OK) return FX_ERR_INVALID;
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(X0 && Xi && W && bias && Xn, "fx_cin_fwd: null pointer");
    CinArgs a;
    memset(&a, 0, sizeof(a));
    a.X0 = X0; a.x0_ld = x0_ld; a.Xi = Xi; a.xi_ld = xi_ld; a.W = W; a.bias = bias; a.Xn = Xn;
    a.pool = pool; a.pool_ld = pool_ld; a.B = B; a.F0 = F0; a.Mi = Mi; a.D = D; a.O = O;
    if (fx_cin_mfma_ok(F0, Mi, D, O)) {
        a.wimg = w_img;
        const int64_t per_wg = 4 * FX_CIN_MFMA_NS;
        const int64_t grid = fx_ceil_div(B, per_wg) < 256 ? fx_ceil_div(B, per_wg) : 256;
        if (Mi <= 16)
            hipLaunchKernelGGL((k_cin_fwd_mfma<4, 40 * 4 * 64>), dim3((unsigned)grid), dim3(256), 0,
                               fx_hip_stream(stream), a);
        else
            hipLaunchKernelGGL((k_cin_fwd_mfma<10, 40 * 10 * 64>), dim3((unsigned)grid), dim3(256), 0,
                               fx_hip_stream(stream), a);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    const int MiP = (Mi + 3) & ~3;
    const size_t lds2 = sizeof(float) * ((size_t)O * F0 * MiP + 4 * (size_t)(F0 + Mi + O) * D);
    if (Mi <= 64 && lds2 <= sizeof(float) * FX_CIN2_LDS_FLOATS) {
        const int64_t grid2 = fx_ceil_div(B, 4) < 256 ? fx_ceil_div(B, 4) : 256;
#define FX_CIN_FWD2(MI)                                                                       \
    hipLaunchKernelGGL(k_cin_fwd2<MI>, dim3((unsigned)grid2), dim3(1024), 0, fx_hip_stream(stream), a)
        if (Mi <= 16) FX_CIN_FWD2(16);
        else if (Mi <= 40) FX_CIN_FWD2(40);
        else FX_CIN_FWD2(64);
#undef FX_CIN_FWD2
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
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
                          float* partial, int64_t B, const float* w_img, fx_stream_t stream) {
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
    hipStream_t s = fx_hip_stream(stream);
    if (fx_cin_mfma_ok(F0, Mi, D, O)) {
        a.wimg = w_img;
        const int64_t per_wg = 4 * FX_CIN_MFMA_NS;
        const int64_t gridx = fx_ceil_div(B, per_wg) < 256 ? fx_ceil_div(B, per_wg) : 256;
        if (Mi <= 16) {
            hipLaunchKernelGGL((k_cin_dx_mfma<1, 40 * 1 * 256>), dim3((unsigned)gridx), dim3(256), 0, s, a);
            hipLaunchKernelGGL((k_cin_dw_mfma<1, 10>), dim3(256), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((k_cin_dx_mfma<3, 40 * 3 * 256>), dim3((unsigned)gridx), dim3(256), 0, s, a);
            hipLaunchKernelGGL((k_cin_dw_mfma<3, 30>), dim3(256), dim3(256), 0, s, a);
        }
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    const int64_t grid = B < 256 ? B : 256;
    const int MiP = (Mi + 3) & ~3;
    const size_t lds_dx2 = sizeof(float) * ((size_t)O * F0 * MiP + 4 * 4 * (size_t)Mi * D);
    if (Mi <= 64 && O <= 32 && D <= 64 && lds_dx2 <= sizeof(float) * FX_CIN2_LDS_FLOATS) {
        const int64_t grid2 = fx_ceil_div(B, 4) < 256 ? fx_ceil_div(B, 4) : 256;
#define FX_CIN_DX2(MI, OM)                                                                    \
    hipLaunchKernelGGL((k_cin_bwd_dx2<MI, OM>), dim3((unsigned)grid2), dim3(1024), 0, s, a)
        if (O <= 16) {
            if (Mi <= 16) FX_CIN_DX2(16, 16);
            else if (Mi <= 40) FX_CIN_DX2(40, 16);
            else FX_CIN_DX2(64, 16);
        } else {
            if (Mi <= 16) FX_CIN_DX2(16, 32);
            else if (Mi <= 40) FX_CIN_DX2(40, 32);
            else FX_CIN_DX2(64, 32);
        }
#undef FX_CIN_DX2
    } else {
        FX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cin_bwd_dx),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dx));
        hipLaunchKernelGGL(k_cin_bwd_dx, dim3((unsigned)grid), dim3(256), lds_dx, s, a);
    }
    // weight-gradient partials always use the full 256 workgroups so `partial` has a fixed shape
    const size_t lds_dw2 = sizeof(float) * 4 * (((size_t)D * MiP + (size_t)(F0 + O) * (D + 1) + 3) & ~(size_t)3);
    if (Mi <= 64 && (int64_t)O * F0 <= 1024 && lds_dw2 <= sizeof(float) * FX_CIN2_LDS_FLOATS) {
#define FX_CIN_DW2(MI) hipLaunchKernelGGL(k_cin_bwd_dw2<MI>, dim3(256), dim3(1024), 0, s, a)
        if (Mi <= 16) FX_CIN_DW2(16);
        else if (Mi <= 40) FX_CIN_DW2(40);
        else FX_CIN_DW2(64);
#undef FX_CIN_DW2
    } else {
        FX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cin_bwd_dw),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dw));
        hipLaunchKernelGGL(k_cin_bwd_dw, dim3(256), dim3(256), lds_dw, s, a);
    }
    FX_CHECK_LAUNCH();
    return FX_OK;
}
