// fx_din.hip — DIN target attention and the Dice activation, first native version:
// the attention MLP runs on the fp32 MFMA GEMM, everything around it is here.
//
// Reference (paths relative to the reference checkout):
//   fuxictr/pytorch/layers/attentions/target_attention.py:66-92   DIN_Attention.forward
//       x_{b,l} = [q_b, k_{b,l}, q_b - k_{b,l}, q_b * k_{b,l}]  -> MLP(4E -> H Dice -> 1) -> * mask
//       -> out_b = sum_l w_{b,l} k_{b,l}
//   fuxictr/pytorch/layers/activations.py:24-51                    Dice
//       p = sigmoid(BatchNorm1d(z; affine=False, eps=1e-9, momentum=0.01)); y = p z + alpha (1-p) z
//       NB the batch statistics run over ALL B*L rows, padded positions included (the mask is
//       applied after the MLP) — reproduced as is.
// All reductions are two-stage with a fixed order (deterministic).
#include "fx_common.h"

#define FX_STAT_CHUNKS 1024

// ---------------------------------------------------------------------------------------------
// attention input [B*L, 4E] and its backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_din_concat_fwd(const float* q, int64_t q_ld,
                                                        const float* K, int64_t k_ldb,
                                                        int64_t k_ldl, int L, int E, int64_t n,
                                                        float* out) {
    // one thread per (row = b*L + l, e)
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int64_t row = i / E;
        const int e = (int)(i - row * E);
        const int64_t b = row / L;
        const int l = (int)(row - b * L);
        const float qv = q[b * q_ld + e];
        const float kv = K[b * k_ldb + (int64_t)l * k_ldl + e];
        float* o = out + row * 4 * E;
        o[e] = qv;
        o[E + e] = kv;
        o[2 * E + e] = qv - kv;
        o[3 * E + e] = qv * kv;
    }
}

// dK[b,l,e] = dx_k - dx_d + dx_p * q ;  dq[b,e] = sum_l (dx_q + dx_d + dx_p * k)  (16 lanes? no:
// one thread per (b,e) loops over l for dq — L is a padded max_len (50), the loop is short)
__global__ __launch_bounds__(256) void k_din_concat_bwd(const float* dx, const float* q,
                                                        int64_t q_ld, const float* K,
                                                        int64_t k_ldb, int64_t k_ldl, int L, int E,
                                                        int64_t B, float* dq, float* dK,
                                                        int64_t dk_ldb, int64_t dk_ldl,
                                                        int accumulate_dk) {
    const int64_t n = B * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / E;
        const int e = (int)(i - b * E);
        const float qv = q[b * q_ld + e];
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const float* d = dx + (b * L + l) * 4 * E;
            const float kv = K[b * k_ldb + (int64_t)l * k_ldl + e];
            const float dxq = d[e], dxk = d[E + e], dxd = d[2 * E + e], dxp = d[3 * E + e];
            acc += dxq + dxd + dxp * kv;
            float* o = dK + b * dk_ldb + (int64_t)l * dk_ldl + e;
            const float g = dxk - dxd + dxp * qv;
            *o = accumulate_dk ? *o + g : g;
        }
        dq[i] = acc;
    }
}

extern "C" int fx_din_concat_fwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                                 int64_t k_ldl, int64_t B, int32_t L, int32_t E, float* out,
                                 fx_stream_t stream) {
    FX_CHECK_ARG(L >= 1 && E >= 1 && B >= 0, "fx_din_concat_fwd: bad sizes");
    if (B == 0) return FX_OK;
    FX_CHECK_ARG(q && K && out, "fx_din_concat_fwd: null pointer");
    const int64_t n = B * L * E;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_din_concat_fwd, dim3((unsigned)blocks), dim3(256), 0,
                       fx_hip_stream(stream), q, q_ld, K, k_ldb, k_ldl, (int)L, (int)E, n, out);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_concat_bwd(const float* dx, const float* q, int64_t q_ld, const float* K,
                                 int64_t k_ldb, int64_t k_ldl, int64_t B, int32_t L, int32_t E,
                                 float* dq, float* dK, int64_t dk_ldb, int64_t dk_ldl,
                                 int32_t accumulate_dk, fx_stream_t stream) {
    FX_CHECK_ARG(L >= 1 && E >= 1 && B >= 0, "fx_din_concat_bwd: bad sizes");
    if (B == 0) return FX_OK;
    FX_CHECK_ARG(dx && q && K && dq && dK, "fx_din_concat_bwd: null pointer");
    int64_t blocks = fx_ceil_div(B * E, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_din_concat_bwd, dim3((unsigned)blocks), dim3(256), 0,
                       fx_hip_stream(stream), dx, q, q_ld, K, k_ldb, k_ldl, (int)L, (int)E, B, dq,
                       dK, dk_ldb, dk_ldl, (int)accumulate_dk);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// masked weighted sum over the sequence and its backward
//   out[b,e] = sum_l w[b,l] * mask[b,l] * K[b,l,e]
//   dw[b,l]  = mask[b,l] * sum_e dout[b,e] K[b,l,e] ;  dK[b,l,e] (+)= w[b,l] mask[b,l] dout[b,e]
// mask is given through the raw id column (id != 0), like `X[seq_field].long() != 0` (DIN.py:125)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_din_pool_fwd(const float* w, const int32_t* ids,
                                                      int64_t ids_ld, const float* K,
                                                      int64_t k_ldb, int64_t k_ldl, int L, int E,
                                                      int64_t B, float* out) {
    const int64_t n = B * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / E;
        const int e = (int)(i - b * E);
        float acc = 0.f;
        for (int l = 0; l < L; ++l) {
            const float m = ids[b * ids_ld + l] != 0 ? 1.f : 0.f;
            acc += (w[b * L + l] * m) * K[b * k_ldb + (int64_t)l * k_ldl + e];
        }
        out[i] = acc;
    }
}

__global__ __launch_bounds__(256) void k_din_pool_bwd(const float* w, const int32_t* ids,
                                                      int64_t ids_ld, const float* K,
                                                      int64_t k_ldb, int64_t k_ldl,
                                                      const float* dout, int L, int E, int Ep,
                                                      int64_t B, float* dw, float* dK,
                                                      int64_t dk_ldb, int64_t dk_ldl) {
    // Ep (power of two >= E) lanes per (b,l)
    const int sub = threadIdx.x & (Ep - 1);
    const int64_t per_block = 256 / Ep;
    const int64_t n = B * L;
    const int64_t n_iter = (n + per_block * gridDim.x - 1) / (per_block * gridDim.x);
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t row = (it * gridDim.x + blockIdx.x) * per_block + threadIdx.x / Ep;
        const bool valid = row < n;
        float dot = 0.f;
        if (valid && sub < E) {
            const int64_t b = row / L;
            const int l = (int)(row - b * L);
            const float m = ids[b * ids_ld + l] != 0 ? 1.f : 0.f;
            const float kv = K[b * k_ldb + (int64_t)l * k_ldl + sub];
            const float dv = dout[b * E + sub];
            dot = m * dv * kv;
            dK[b * dk_ldb + (int64_t)l * dk_ldl + sub] = (w[row] * m) * dv;
        }
        for (int off = 1; off < Ep; off <<= 1) dot += __shfl_xor(dot, off, 64);
        if (valid && sub == 0) dw[row] = dot;
    }
}

extern "C" int fx_din_pool_fwd(const float* w, const int32_t* ids, int64_t ids_ld, const float* K,
                               int64_t k_ldb, int64_t k_ldl, int64_t B, int32_t L, int32_t E,
                               float* out, fx_stream_t stream) {
    FX_CHECK_ARG(L >= 1 && E >= 1 && B >= 0, "fx_din_pool_fwd: bad sizes");
    if (B == 0) return FX_OK;
    FX_CHECK_ARG(w && ids && K && out, "fx_din_pool_fwd: null pointer");
    int64_t blocks = fx_ceil_div(B * E, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_din_pool_fwd, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream),
                       w, ids, ids_ld, K, k_ldb, k_ldl, (int)L, (int)E, B, out);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_pool_bwd(const float* w, const int32_t* ids, int64_t ids_ld, const float* K,
                               int64_t k_ldb, int64_t k_ldl, const float* dout, int64_t B,
                               int32_t L, int32_t E, float* dw, float* dK, int64_t dk_ldb,
                               int64_t dk_ldl, fx_stream_t stream) {
    FX_CHECK_ARG(L >= 1 && E >= 1 && E <= 64 && B >= 0, "fx_din_pool_bwd: bad sizes (E <= 64)");
    if (B == 0) return FX_OK;
    FX_CHECK_ARG(w && ids && K && dout && dw && dK, "fx_din_pool_bwd: null pointer");
    int Ep = 1;
    while (Ep < E) Ep <<= 1;
    int64_t blocks = fx_ceil_div(B * L, 256 / Ep);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_din_pool_bwd, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream),
                       w, ids, ids_ld, K, k_ldb, k_ldl, dout, (int)L, (int)E, Ep, B, dw, dK, dk_ldb,
                       dk_ldl);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// Dice.  stats[0..H) = mean, stats[H..2H) = biased variance of the batch (training) or the running
// statistics (eval).  Two-stage column reductions over N rows.
// ---------------------------------------------------------------------------------------------
// stage 1: partial[c][k][h] = sum over rows of chunk c of term k (k < NT)
template <int MODE>  // 0: (z, z^2)   1: backward sums (dalpha, dzhat, dzhat*zhat)
__global__ __launch_bounds__(256) void k_dice_reduce(const float* Z, const float* dY,
                                                     const float* stats, const float* alpha,
                                                     float eps, int64_t N, int H, int64_t rows,
                                                     float* partial) {
    constexpr int NT = MODE == 0 ? 2 : 3;
    __shared__ float red[NT][256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t h = (int64_t)blockIdx.x * 64 + tx;
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    const int64_t r1 = (r0 + rows < N) ? r0 + rows : N;
    float acc[NT];
#pragma unroll
    for (int k = 0; k < NT; ++k) acc[k] = 0.f;
    if (h < H) {
        float mean = 0.f, rstd = 0.f, al = 0.f;
        if (MODE == 1) {
            mean = stats[h];
            rstd = rsqrtf(stats[H + h] + eps);
            al = alpha[h];
        }
        for (int64_t r = r0 + ty; r < r1; r += 4) {
            const float z = Z[r * H + h];
            if (MODE == 0) {
                acc[0] += z;
                acc[1] = fmaf(z, z, acc[1]);
            } else {
                const float zh = (z - mean) * rstd;
                const float p = 1.f / (1.f + expf(-zh));
                const float dy = dY[r * H + h];
                const float dzh = dy * z * (1.f - al) * p * (1.f - p);
                acc[0] = fmaf(dy * (1.f - p), z, acc[0]);   // d alpha
                acc[1] += dzh;
                acc[2] = fmaf(dzh, zh, acc[2]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    if (ty == 0 && h < H) {
#pragma unroll
        for (int k = 0; k < NT; ++k)
            partial[((int64_t)blockIdx.y * NT + k) * H + h] =
                (red[k][tx] + red[k][tx + 64]) + (red[k][tx + 128] + red[k][tx + 192]);
    }
}

// H % 4 == 0: float4 columns, 16 threads per 64-column row segment, 16 row lanes, rows unrolled x2:
// the statistics pass is a pure HBM stream (52 MB at B*L = 204800, H = 64) and needs many loads in
// flight per CU to reach the bandwidth the one-float-per-thread version (above) cannot.
template <int MODE>
__global__ __launch_bounds__(256) void k_dice_reduce_v4(const float* Z, const float* dY,
                                                        const float* stats, const float* alpha,
                                                        float eps, int64_t N, int H, int64_t rows,
                                                        float* partial) {
    constexpr int NT = MODE == 0 ? 2 : 3;
    __shared__ float red[NT][16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t h = (int64_t)blockIdx.x * 64 + tx * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    const int64_t r1 = (r0 + rows < N) ? r0 + rows : N;
    float acc[NT][4];
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[k][e] = 0.f;
    if (h < H) {
        float mean[4] = {0.f, 0.f, 0.f, 0.f}, rstd[4] = {0.f, 0.f, 0.f, 0.f}, al[4] = {0.f, 0.f, 0.f, 0.f};
        if (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                mean[e] = stats[h + e];
                rstd[e] = rsqrtf(stats[H + h + e] + eps);
                al[e] = alpha[h + e];
            }
        }
        auto term = [&](const float4& zq, const float4& dq) {
            const float z[4] = {zq.x, zq.y, zq.z, zq.w};
            const float d[4] = {dq.x, dq.y, dq.z, dq.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (MODE == 0) {
                    acc[0][e] += z[e];
                    acc[1][e] = fmaf(z[e], z[e], acc[1][e]);
                } else {
                    const float zh = (z[e] - mean[e]) * rstd[e];
                    const float pr = 1.f / (1.f + expf(-zh));
                    const float dzh = d[e] * z[e] * (1.f - al[e]) * pr * (1.f - pr);
                    acc[0][e] = fmaf(d[e] * (1.f - pr), z[e], acc[0][e]);
                    acc[1][e] += dzh;
                    acc[2][e] = fmaf(dzh, zh, acc[2][e]);
                }
            }
        };
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
        int64_t r = r0 + ty;
        for (; r + 16 < r1; r += 32) {
            const float4 z0 = *reinterpret_cast<const float4*>(Z + r * H + h);
            const float4 z1 = *reinterpret_cast<const float4*>(Z + (r + 16) * H + h);
            float4 d0 = zero, d1 = zero;
            if (MODE == 1) {
                d0 = *reinterpret_cast<const float4*>(dY + r * H + h);
                d1 = *reinterpret_cast<const float4*>(dY + (r + 16) * H + h);
            }
            term(z0, d0);
            term(z1, d1);
        }
        for (; r < r1; r += 16) {
            const float4 z0 = *reinterpret_cast<const float4*>(Z + r * H + h);
            float4 d0 = zero;
            if (MODE == 1) d0 = *reinterpret_cast<const float4*>(dY + r * H + h);
            term(z0, d0);
        }
    }
#pragma unroll
    for (int k = 0; k < NT; ++k)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[k][ty][tx * 4 + e] = acc[k][e];
    __syncthreads();
    if (threadIdx.x < 64) {
        const int64_t hh = (int64_t)blockIdx.x * 64 + threadIdx.x;
        if (hh < H) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                float sum = 0.f;
#pragma unroll
                for (int y = 0; y < 16; ++y) sum += red[k][y][threadIdx.x];
                partial[((int64_t)blockIdx.y * NT + k) * H + hh] = sum;
            }
        }
    }
}

// fixed-order sum over the chunks of one term: 64 columns x 4 chunk lanes per workgroup
// sum over the chunks of term k for 16 columns per workgroup: 16 chunk lanes, 8 independent loads
// in flight per lane, fixed order -> deterministic.  Returns the column sum to every lane of the
// column; `red` is [16][16].
__device__ __forceinline__ float fx_chunk_sum(const float* partial, int nt, int k, int chunks,
                                              int H, int h, int ty, float (*red)[16]) {
    float s = 0.f;
    if (h < H) {
        int c = ty;
        for (; c + 7 * 16 < chunks; c += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((int64_t)(c + u * 16) * nt + k) * H + h];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < chunks; c += 16) s += partial[((int64_t)c * nt + k) * H + h];
    }
    const int tx = threadIdx.x & 15;
    __syncthreads();
    red[ty][tx] = s;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int y = 0; y < 16; ++y) t += red[y][tx];
    return t;
}
__global__ __launch_bounds__(256) void k_dice_stats_final(const float* partial, int chunks, int H,
                                                          int64_t N, float momentum, float* stats,
                                                          float* running_mean,
                                                          float* running_var) {
    __shared__ float red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int h = blockIdx.x * 16 + tx;
    const double s = (double)fx_chunk_sum(partial, 2, 0, chunks, H, h, ty, red);
    const double ss = (double)fx_chunk_sum(partial, 2, 1, chunks, H, h, ty, red);
    if (ty != 0 || h >= H) return;
    const double mean = s / (double)N;
    double var = ss / (double)N - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[h] = (float)mean;
    stats[H + h] = (float)var;
    if (running_mean) {
        const double unb = N > 1 ? var * (double)N / (double)(N - 1) : var;
        running_mean[h] = (float)((1.0 - momentum) * running_mean[h] + momentum * mean);
        running_var[h] = (float)((1.0 - momentum) * running_var[h] + momentum * unb);
    }
}

__global__ __launch_bounds__(256) void k_dice_bwd_final(const float* partial, int chunks, int H,
                                                        float* sums) {
    __shared__ float red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int h = blockIdx.x * 16 + tx;
    for (int k = 0; k < 3; ++k) {
        const float s = fx_chunk_sum(partial, 3, k, chunks, H, h, ty, red);
        if (ty == 0 && h < H) sums[k * H + h] = s;
    }
}

// forward apply: y = z (p + alpha (1 - p)),  p = sigmoid((z - mean) rstd)
__global__ __launch_bounds__(256) void k_dice_fwd(const float* Z, const float* stats,
                                                  const float* alpha, float eps, int64_t n, int H,
                                                  float* Y) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int h = (int)(i % H);
        const float z = Z[i];
        const float zh = (z - stats[h]) * rsqrtf(stats[H + h] + eps);
        const float p = 1.f / (1.f + expf(-zh));
        Y[i] = p * z + alpha[h] * (1.f - p) * z;
    }
}

// backward apply: dz = dy (p + alpha(1-p)) + rstd (dzhat - [mean(dzhat) + zhat mean(dzhat zhat)])
// (the bracket only in training mode, where the statistics depend on z)
__global__ __launch_bounds__(256) void k_dice_bwd(const float* Z, const float* dY,
                                                  const float* stats, const float* alpha,
                                                  const float* sums, float eps, int64_t n, int H,
                                                  int64_t N, int training, float* dZ) {
    const float invN = 1.f / (float)N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * 256) {
        const int h = (int)(i % H);
        const float z = Z[i], dy = dY[i], al = alpha[h];
        const float rstd = rsqrtf(stats[H + h] + eps);
        const float zh = (z - stats[h]) * rstd;
        const float p = 1.f / (1.f + expf(-zh));
        float dzh = dy * z * (1.f - al) * p * (1.f - p);
        if (training) dzh -= sums[H + h] * invN + zh * (sums[2 * H + h] * invN);
        dZ[i] = dy * (p + al * (1.f - p)) + dzh * rstd;
    }
}

extern "C" int64_t fx_dice_workspace_floats(int32_t H) { return (int64_t)FX_STAT_CHUNKS * 3 * H; }

extern "C" int fx_dice_fwd(const float* Z, int64_t N, int32_t H, const float* alpha, float eps,
                           float momentum, int32_t training, float* running_mean,
                           float* running_var, float* stats, float* Y, float* workspace,
                           fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1, "fx_dice_fwd: bad sizes");
    FX_CHECK_ARG(Z && alpha && stats && Y && running_mean && running_var,
                 "fx_dice_fwd: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    if (training) {
        FX_CHECK_ARG(workspace, "fx_dice_fwd: training mode needs a workspace");
        const int64_t rows = fx_ceil_div(N, FX_STAT_CHUNKS);
        if (H % 4 == 0 && (reinterpret_cast<uintptr_t>(Z) & 15) == 0)
            hipLaunchKernelGGL(k_dice_reduce_v4<0>, dim3((unsigned)fx_ceil_div(H, 64), FX_STAT_CHUNKS),
                               dim3(256), 0, s, Z, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, eps, N, (int)H, rows, workspace);
        else
            hipLaunchKernelGGL(k_dice_reduce<0>, dim3((unsigned)fx_ceil_div(H, 64), FX_STAT_CHUNKS),
                               dim3(256), 0, s, Z, (const float*)nullptr, (const float*)nullptr,
                               (const float*)nullptr, eps, N, (int)H, rows, workspace);
        hipLaunchKernelGGL(k_dice_stats_final, dim3((unsigned)fx_ceil_div(H, 16)), dim3(256), 0, s,
                           workspace, (int)FX_STAT_CHUNKS, (int)H, N, momentum, stats, running_mean,
                           running_var);
    } else {
        // eval: normalise with the running statistics (BatchNorm1d.eval())
        FX_CHECK_HIP(hipMemcpyAsync(stats, running_mean, sizeof(float) * H,
                                    hipMemcpyDeviceToDevice, s));
        FX_CHECK_HIP(hipMemcpyAsync(stats + H, running_var, sizeof(float) * H,
                                    hipMemcpyDeviceToDevice, s));
    }
    const int64_t n = N * H;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_dice_fwd, dim3((unsigned)blocks), dim3(256), 0, s, Z, stats, alpha, eps, n,
                       (int)H, Y);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_bwd(const float* Z, const float* dY, int64_t N, int32_t H,
                           const float* alpha, float eps, int32_t training, const float* stats,
                           float* dZ, float* dalpha, float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1, "fx_dice_bwd: bad sizes");
    FX_CHECK_ARG(Z && dY && alpha && stats && dZ && dalpha && workspace,
                 "fx_dice_bwd: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    const int64_t rows = fx_ceil_div(N, FX_STAT_CHUNKS);
    float* sums = workspace + (int64_t)FX_STAT_CHUNKS * 3 * H - 3 * H;  // tail of the workspace
    // partials occupy [0, (CHUNKS-1)*3*H)?  no: keep them disjoint — use CHUNKS-1 chunks of rows
    const int chunks = FX_STAT_CHUNKS - 1;
    const int64_t rows2 = fx_ceil_div(N, chunks);
    (void)rows;
    if (H % 4 == 0 && ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(dY)) & 15) == 0)
        hipLaunchKernelGGL(k_dice_reduce_v4<1>, dim3((unsigned)fx_ceil_div(H, 64), chunks), dim3(256),
                           0, s, Z, dY, stats, alpha, eps, N, (int)H, rows2, workspace);
    else
        hipLaunchKernelGGL(k_dice_reduce<1>, dim3((unsigned)fx_ceil_div(H, 64), chunks), dim3(256), 0,
                           s, Z, dY, stats, alpha, eps, N, (int)H, rows2, workspace);
    hipLaunchKernelGGL(k_dice_bwd_final, dim3((unsigned)fx_ceil_div(H, 16)), dim3(256), 0, s,
                       workspace, chunks, (int)H, sums);
    FX_CHECK_HIP(hipMemcpyAsync(dalpha, sums, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
    const int64_t n = N * H;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_dice_bwd, dim3((unsigned)blocks), dim3(256), 0, s, Z, dY, stats, alpha,
                       sums, eps, n, (int)H, N, (int)training, dZ);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// Dice across ranks (row-sharded training: every rank holds B/N samples of the global batch).  The
// reference normalises with the statistics of the WHOLE batch (activations.py:40-51), so the two
// column reductions are split from their consumers: local sums -> (the host all-reduces them) ->
// apply with the global row count.  Same kernels as fx_dice_fwd / fx_dice_bwd.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_dice_sums_final(const float* partial, int chunks, int H,
                                                         int nt, float* sums) {
    __shared__ float red[16][16];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int h = blockIdx.x * 16 + tx;
    for (int k = 0; k < nt; ++k) {
        const float s = fx_chunk_sum(partial, nt, k, chunks, H, h, ty, red);
        if (ty == 0 && h < H) sums[k * H + h] = s;
    }
}

__global__ __launch_bounds__(256) void k_dice_stats_from_sums(const float* sums, int H, double n_total,
                                                              float momentum, float* stats,
                                                              float* running_mean,
                                                              float* running_var) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h >= H) return;
    const double mean = (double)sums[h] / n_total;
    double var = (double)sums[H + h] / n_total - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[h] = (float)mean;
    stats[H + h] = (float)var;
    if (running_mean) {
        const double unb = n_total > 1.0 ? var * n_total / (n_total - 1.0) : var;
        running_mean[h] = (float)((1.0 - momentum) * running_mean[h] + momentum * mean);
        running_var[h] = (float)((1.0 - momentum) * running_var[h] + momentum * unb);
    }
}

extern "C" int fx_dice_local_sums(const float* Z, int64_t N, int32_t H, float* sums,
                                  float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1, "fx_dice_local_sums: bad sizes");
    FX_CHECK_ARG(Z && sums && workspace, "fx_dice_local_sums: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    const int64_t rows = fx_ceil_div(N, FX_STAT_CHUNKS);
    if (H % 4 == 0 && (reinterpret_cast<uintptr_t>(Z) & 15) == 0)
        hipLaunchKernelGGL(k_dice_reduce_v4<0>, dim3((unsigned)fx_ceil_div(H, 64), FX_STAT_CHUNKS),
                           dim3(256), 0, s, Z, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, 0.f, N, (int)H, rows, workspace);
    else
        hipLaunchKernelGGL(k_dice_reduce<0>, dim3((unsigned)fx_ceil_div(H, 64), FX_STAT_CHUNKS),
                           dim3(256), 0, s, Z, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, 0.f, N, (int)H, rows, workspace);
    hipLaunchKernelGGL(k_dice_sums_final, dim3((unsigned)fx_ceil_div(H, 16)), dim3(256), 0, s,
                       workspace, (int)FX_STAT_CHUNKS, (int)H, 2, sums);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_fwd_from_sums(const float* Z, int64_t N, int32_t H, const float* alpha,
                                     float eps, float momentum, const float* sums, int64_t n_total,
                                     float* running_mean, float* running_var, float* stats,
                                     float* Y, fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1 && n_total >= N, "fx_dice_fwd_from_sums: bad sizes");
    FX_CHECK_ARG(Z && alpha && sums && stats && Y, "fx_dice_fwd_from_sums: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    hipLaunchKernelGGL(k_dice_stats_from_sums, dim3((unsigned)fx_ceil_div(H, 256)), dim3(256), 0, s,
                       sums, (int)H, (double)n_total, momentum, stats, running_mean, running_var);
    const int64_t n = N * H;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_dice_fwd, dim3((unsigned)blocks), dim3(256), 0, s, Z, stats, alpha, eps, n,
                       (int)H, Y);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_bwd_local_sums(const float* Z, const float* dY, int64_t N, int32_t H,
                                      const float* alpha, float eps, const float* stats,
                                      float* sums3, float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1, "fx_dice_bwd_local_sums: bad sizes");
    FX_CHECK_ARG(Z && dY && alpha && stats && sums3 && workspace,
                 "fx_dice_bwd_local_sums: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    const int chunks = FX_STAT_CHUNKS - 1;
    const int64_t rows2 = fx_ceil_div(N, chunks);
    if (H % 4 == 0 && ((reinterpret_cast<uintptr_t>(Z) | reinterpret_cast<uintptr_t>(dY)) & 15) == 0)
        hipLaunchKernelGGL(k_dice_reduce_v4<1>, dim3((unsigned)fx_ceil_div(H, 64), chunks), dim3(256),
                           0, s, Z, dY, stats, alpha, eps, N, (int)H, rows2, workspace);
    else
        hipLaunchKernelGGL(k_dice_reduce<1>, dim3((unsigned)fx_ceil_div(H, 64), chunks), dim3(256), 0,
                           s, Z, dY, stats, alpha, eps, N, (int)H, rows2, workspace);
    hipLaunchKernelGGL(k_dice_sums_final, dim3((unsigned)fx_ceil_div(H, 16)), dim3(256), 0, s,
                       workspace, chunks, (int)H, 3, sums3);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_bwd_from_sums(const float* Z, const float* dY, int64_t N, int32_t H,
                                     const float* alpha, float eps, const float* stats,
                                     const float* sums3, int64_t n_total, float* dZ,
                                     fx_stream_t stream) {
    FX_CHECK_ARG(N >= 1 && H >= 1 && n_total >= N, "fx_dice_bwd_from_sums: bad sizes");
    FX_CHECK_ARG(Z && dY && alpha && stats && sums3 && dZ, "fx_dice_bwd_from_sums: null pointer");
    const int64_t n = N * H;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_dice_bwd, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), Z, dY,
                       stats, alpha, sums3, eps, n, (int)H, n_total, 1, dZ);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
