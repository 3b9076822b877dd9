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
// fixed order.  These are the fp32 VALU kernels: they take every shape; D = 16 with O <= 16 and
// F0, Mi <= 40 (the BASELINE xDeepFM) runs on the matrix cores instead (fx_cin_mfma.hip).
#include "fx_cin.h"

#include <stdlib.h>

#define FX_CIN_MAX_W_FLOATS (30 * 1024)   // 120 KB of W (or dW) per workgroup
#define FX_CIN_MAX_TILE 4096              // F0*D and Mi*D and O*D staged per sample


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
    float* out = a.partial + (int64_t)blockIdx.x * a.partial_ld;
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
    float* out = a.partial + (int64_t)blockIdx.x * a.partial_ld;
    if (own) {
#pragma unroll
        for (int m = 0; m < MI; ++m)
            if (m < Mi) out[(o * F0 + h) * Mi + m] = acc[m];
        if (h == 0) out[O * C + o] = bacc;
    }
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
    return fx_cin_mfma_shape(F0, Mi, D, O) ? fx_cin_mfma_wimg_floats(F0, Mi) : 0;
}

extern "C" int fx_cin_pack_w(int32_t n_layers, const float* const* W, const int32_t* F0, const int32_t* Mi,
                             int32_t D, const int32_t* O, float* const* w_img, fx_stream_t stream) {
    FX_CHECK_ARG(n_layers >= 1 && n_layers <= FX_CIN_PACK_MAX, "fx_cin_pack_w: 1..4 layers per call");
    FX_CHECK_ARG(W && F0 && Mi && O && w_img, "fx_cin_pack_w: null pointer");
    CinPackArgs pa;
    memset(&pa, 0, sizeof(pa));
    pa.n = n_layers;
    for (int i = 0; i < n_layers; ++i) {
        FX_CHECK_ARG(W[i] && w_img[i], "fx_cin_pack_w: null pointer");
        FX_CHECK_ARG(((uintptr_t)w_img[i] & 15) == 0, "fx_cin_pack_w: w_img must be 16-byte aligned");
        FX_CHECK_ARG(fx_cin_mfma_shape(F0[i], Mi[i], D, O[i]),
                     "fx_cin_pack_w: no image for this shape (fx_cin_wimg_floats == 0)");
        pa.W[i] = W[i]; pa.img[i] = w_img[i]; pa.F0[i] = F0[i]; pa.Mi[i] = Mi[i]; pa.O[i] = O[i];
    }
    fx_cin_mfma_pack_w(pa, fx_hip_stream(stream));
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
    if (fx_cin_mfma_shape(F0, Mi, D, O)) {
        a.wimg = w_img;
        if (fx_cin_mfma_fwd(a, fx_hip_stream(stream))) {
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
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
                          float* partial, int64_t partial_ld, int64_t B, const float* w_img,
                          fx_stream_t stream) {
    if (fx_cin_check("fx_cin_bwd", F0, Mi, D, O) != FX_OK) return FX_ERR_INVALID;
    FX_CHECK_ARG(B >= 1, "fx_cin_bwd: B must be >= 1");
    FX_CHECK_ARG(X0 && Xi && W && (dXn || dpool) && dX0 && dXi && partial,
                 "fx_cin_bwd: null pointer");
    FX_CHECK_ARG(partial_ld >= (int64_t)O * F0 * Mi + O, "fx_cin_bwd: partial_ld < O*F0*Mi + O");
    CinArgs a;
    memset(&a, 0, sizeof(a));
    a.X0 = X0; a.x0_ld = x0_ld; a.Xi = Xi; a.xi_ld = xi_ld; a.W = W; a.dXn = dXn; a.dpool = dpool;
    a.dpool_ld = dpool_ld; a.dX0 = dX0; a.dx0_ld = dx0_ld; a.acc_dx0 = accumulate_dx0; a.dXi = dXi;
    a.dxi_ld = dxi_ld; a.partial = partial; a.partial_ld = partial_ld; a.B = B; a.F0 = F0; a.Mi = Mi; a.D = D; a.O = O;
    int Dp = 1;
    while (Dp < D) Dp <<= 1;
    const int NH = 256 / Dp;
    const size_t lds_dx = sizeof(float) * ((size_t)O * F0 * Mi + (size_t)(F0 + Mi + O) * D +
                                           (size_t)NH * Mi * D);
    const size_t lds_dw = sizeof(float) * ((size_t)O * F0 * Mi + O + (size_t)(F0 + Mi + O) * D);
    FX_CHECK_ARG(lds_dx <= 160 * 1024 && lds_dw <= 160 * 1024, "fx_cin_bwd: LDS need too large");
    hipStream_t s = fx_hip_stream(stream);
    if (fx_cin_mfma_shape(F0, Mi, D, O)) {
        a.wimg = w_img;
        if (fx_cin_mfma_bwd(a, s)) {
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
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
