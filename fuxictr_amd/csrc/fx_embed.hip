// fx_embed.hip — input packing, multi-field embedding gather, numeric-weight gradient,
// FM second-order term and LR first-order term.  HBM/latency-bound kernels: one 64-byte row
// (D=16 fp32) is read by a quad of lanes as 4 x float4, lookups of one sample are adjacent so
// the [B,F,D] record is written as whole 128-B lines, no intermediate per-field tensors exist.
//
// Reference behaviour restated here (paths relative to the reference checkout):
//   fuxictr/pytorch/layers/embeddings/feature_embedding.py:261-297 (per-feature lookup loop),
//   :230-259 (stack/cat), fuxictr/pytorch/layers/blocks/logistic_regression.py:55-58,
//   fuxictr/pytorch/layers/interactions/inner_product.py:55-62.
#include "fx_common.h"

#include <stdlib.h>

// ---------------------------------------------------------------------------------------------
// fx_pack_columns
// ---------------------------------------------------------------------------------------------
#define FX_PACK_MAX_COLS 64
struct PackArgs {
    const void* col[FX_PACK_MAX_COLS];
    int32_t dtype[FX_PACK_MAX_COLS];
    int32_t width[FX_PACK_MAX_COLS];
    int64_t out_col[FX_PACK_MAX_COLS];
    int64_t B;
    int64_t out_ld;
    void* out;
};

template <typename OutT>
__global__ __launch_bounds__(256) void k_pack_columns(PackArgs a) {
    const int c = blockIdx.y;
    const int64_t w = a.width[c];
    const int64_t n = a.B * w;
    const int dt = a.dtype[c];
    const void* src = a.col[c];
    OutT* out = reinterpret_cast<OutT*>(a.out);
    const int64_t oc = a.out_col[c];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / w, k = i - b * w;
        OutT v;
        if (dt == FX_F64) v = (OutT) reinterpret_cast<const double*>(src)[i];
        else if (dt == FX_I64) v = (OutT) reinterpret_cast<const int64_t*>(src)[i];
        else if (dt == FX_F32) v = (OutT) reinterpret_cast<const float*>(src)[i];
        else v = (OutT) reinterpret_cast<const int32_t*>(src)[i];
        out[b * a.out_ld + oc + k] = v;
    }
}

extern "C" int fx_pack_columns(const void* const* cols_host, const int32_t* dtypes_host,
                               const int32_t* widths_host, int32_t ncols, int64_t B,
                               int32_t out_dtype, void* out, int64_t out_ld, int64_t out_col0,
                               fx_stream_t stream) {
    FX_CHECK_ARG(ncols >= 0 && ncols <= FX_PACK_MAX_COLS, "fx_pack_columns: ncols=%d not in [0,%d]",
                 ncols, FX_PACK_MAX_COLS);
    FX_CHECK_ARG(out_dtype == FX_I32 || out_dtype == FX_F32,
                 "fx_pack_columns: out_dtype must be FX_I32 or FX_F32");
    FX_CHECK_ARG(B >= 0, "fx_pack_columns: B < 0");
    if (ncols == 0 || B == 0) return FX_OK;
    FX_CHECK_ARG(out != nullptr && cols_host && dtypes_host && widths_host,
                 "fx_pack_columns: null pointer");
    PackArgs a;
    memset(&a, 0, sizeof(a));
    int64_t col = out_col0, maxw = 1;
    for (int c = 0; c < ncols; ++c) {
        FX_CHECK_ARG(cols_host[c] != nullptr, "fx_pack_columns: column %d is null", c);
        FX_CHECK_ARG(dtypes_host[c] >= FX_F32 && dtypes_host[c] <= FX_I64,
                     "fx_pack_columns: bad dtype %d for column %d", dtypes_host[c], c);
        FX_CHECK_ARG(widths_host[c] >= 1, "fx_pack_columns: width of column %d < 1", c);
        a.col[c] = cols_host[c];
        a.dtype[c] = dtypes_host[c];
        a.width[c] = widths_host[c];
        a.out_col[c] = col;
        col += widths_host[c];
        if (widths_host[c] > maxw) maxw = widths_host[c];
    }
    FX_CHECK_ARG(col <= out_ld, "fx_pack_columns: columns (%lld) exceed out_ld (%lld)",
                 (long long)col, (long long)out_ld);
    a.B = B;
    a.out_ld = out_ld;
    a.out = out;
    int64_t gx = fx_ceil_div(B * maxw, 256);
    if (gx > 1024) gx = 1024;
    dim3 grid((unsigned)gx, (unsigned)ncols);
    if (out_dtype == FX_I32)
        hipLaunchKernelGGL(k_pack_columns<int32_t>, grid, dim3(256), 0, fx_hip_stream(stream), a);
    else
        hipLaunchKernelGGL(k_pack_columns<float>, grid, dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_gather_fwd: item i -> (b = i / (C+Fd), r = i % (C+Fd)); r < C is an id lookup, else a
// numeric expansion.  `lanes` lanes (VEC floats each) serve one item.
// ---------------------------------------------------------------------------------------------
struct GatherArgs {
    const float* table;
    const int32_t* ids;
    int64_t ids_ld;
    const int64_t* col_row_base;
    const int32_t* col_vocab;
    const int64_t* col_out_off;
    const float* dense;
    int64_t dense_ld;
    const float* num_w;
    const int64_t* num_out_off;
    float* out;
    int64_t out_ld;
    int64_t B;
    fx_scalars* scal;
    int32_t D, C, Fd, lanes_log2;
    int64_t table_ld;     // row stride of `table` in floats (D for a packed table, W inside a row record)
};

template <int VEC>
__global__ __launch_bounds__(256) void k_emb_gather_fwd(GatherArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int64_t items_per_block = 256 >> a.lanes_log2;
    const int R = a.C + a.Fd;
    const int64_t n_items = a.B * R;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    for (int64_t item = (int64_t)blockIdx.x * items_per_block + (threadIdx.x >> a.lanes_log2);
         item < n_items; item += (int64_t)gridDim.x * items_per_block) {
        const int64_t b = item / R;
        const int r = (int)(item - b * R);
        float val[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) val[k] = 0.f;
        int64_t off;
        if (r < a.C) {
            const int32_t id = a.ids[b * a.ids_ld + r];
            off = a.col_out_off[r];
            if (id >= 0 && id < a.col_vocab[r]) {
                if (lane_on) {
                    const int64_t row = a.col_row_base[r] + id;
                    fx_load<VEC>(a.table + row * a.table_ld + d0, val);
                }
            } else if (sub == 0) {
                atomicOr(&a.scal->err_flag, FX_FLAG_BAD_ID);
            }
        } else {
            const int j = r - a.C;
            off = a.num_out_off[j];
            if (lane_on) {
                const float x = a.dense[b * a.dense_ld + j];
                float w[VEC];
                fx_load<VEC>(a.num_w + (int64_t)j * a.D + d0, w);
#pragma unroll
                for (int k = 0; k < VEC; ++k) val[k] = x * w[k];
            }
        }
        if (lane_on) fx_store<VEC>(a.out + b * a.out_ld + off + d0, val);
    }
}

static int fx_log2i(int x) {
    int l = 0;
    while ((1 << l) < x) ++l;
    return l;
}

extern "C" int fx_emb_gather_fwd(const float* table, int32_t D, const int32_t* ids,
                                 int64_t ids_ld, const int64_t* col_row_base,
                                 const int32_t* col_vocab, const int64_t* col_out_off, int32_t C,
                                 const float* dense, int64_t dense_ld, const float* num_w,
                                 const int64_t* num_out_off, int32_t Fd, float* out,
                                 int64_t out_ld, int64_t B, fx_scalars* scal, int64_t table_ld,
                                 fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_emb_gather_fwd: D=%d not in [1,256]", D);
    if (table_ld <= 0) table_ld = D;
    FX_CHECK_ARG(C >= 0 && Fd >= 0 && B >= 0, "fx_emb_gather_fwd: negative size");
    if (B == 0 || C + Fd == 0) return FX_OK;
    FX_CHECK_ARG(out && scal, "fx_emb_gather_fwd: null out/scal");
    FX_CHECK_ARG(C == 0 || (table && ids && col_row_base && col_vocab && col_out_off),
                 "fx_emb_gather_fwd: null sparse argument");
    FX_CHECK_ARG(Fd == 0 || (dense && num_w && num_out_off),
                 "fx_emb_gather_fwd: null numeric argument");
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(out_ld % g.vec == 0, "fx_emb_gather_fwd: out_ld=%lld not a multiple of %d",
                 (long long)out_ld, g.vec);
    FX_CHECK_ARG(table_ld >= D && table_ld % g.vec == 0,
                 "fx_emb_gather_fwd: table_ld=%lld does not allow %d-wide row loads", (long long)table_ld, g.vec);
    GatherArgs a{table, ids, ids_ld, col_row_base, col_vocab, col_out_off, dense, dense_ld,
                 num_w, num_out_off, out, out_ld, B, scal, D, C, Fd, fx_log2i(g.lanes), table_ld};
    const int64_t items = B * (int64_t)(C + Fd);
    int64_t blocks = fx_ceil_div(items, 256 / g.lanes);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (g.vec == 4) hipLaunchKernelGGL(k_emb_gather_fwd<4>, grid, dim3(256), 0, s, a);
    else if (g.vec == 2) hipLaunchKernelGGL(k_emb_gather_fwd<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_emb_gather_fwd<1>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_seq_pool_fwd: sequence features behind MaskedSumPooling / MaskedAveragePooling
// (fuxictr/pytorch/layers/pooling.py:32-47, :59-70) — the [B, L, D] history is never written: one
// wave per (sample, sequence feature); the wave's 64 lanes are P = 64/lanes position groups of
// `lanes` lanes (VEC floats each), group p takes positions p, p+P, ...; a butterfly over the
// groups combines the partial sums in a fixed order.  A position counts towards the mean when
// its row does not sum to exactly 0 (the reference's mask for mask=None).
// ---------------------------------------------------------------------------------------------
struct SeqPoolArgs {
    const float* table;
    const int32_t* ids;
    int64_t ids_ld;
    const int64_t* col_row_base;
    const int32_t* col_vocab;
    const int32_t* seq_col0;
    const int32_t* seq_len;
    const int32_t* seq_mode;
    const int64_t* seq_out_off;
    float* out;
    int64_t out_ld;
    float* denom;
    int64_t B;
    fx_scalars* scal;
    int32_t D, n_seq, lanes_log2;
    int64_t table_ld;     // row stride of `table` in floats
};

template <int VEC>
__global__ __launch_bounds__(256) void k_emb_seq_pool_fwd(SeqPoolArgs a) {
    const int lane = threadIdx.x & 63;
    const int lanes = 1 << a.lanes_log2;
    const int sub = lane & (lanes - 1);
    const int p = lane >> a.lanes_log2;
    const int P = 64 >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int64_t n_items = a.B * a.n_seq;
    for (int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); item < n_items;
         item += (int64_t)gridDim.x * 4) {                       // wave-uniform
        const int64_t b = item / a.n_seq;
        const int s = (int)(item - b * a.n_seq);
        const int c0 = a.seq_col0[s], L = a.seq_len[s];
        const int32_t* row_ids = a.ids + b * a.ids_ld + c0;
        float acc[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
        float cnt = 0.f;
        for (int l0 = 0; l0 < L; l0 += P) {                      // wave-uniform trip count
            const int l = l0 + p;
            float val[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) val[k] = 0.f;
            if (l < L) {
                const int32_t id = row_ids[l];
                if (id >= 0 && id < a.col_vocab[c0 + l]) {
                    if (lane_on)
                        fx_load<VEC>(a.table + (a.col_row_base[c0 + l] + id) * a.table_ld + d0, val);
                } else if (sub == 0) {
                    atomicOr(&a.scal->err_flag, FX_FLAG_BAD_ID);
                }
            }
            float rs = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) { rs += val[k]; acc[k] += val[k]; }
            for (int off = 1; off < lanes; off <<= 1) rs += __shfl_xor(rs, off, 64);
            cnt += (rs != 0.f) ? 1.f : 0.f;
        }
        for (int off = lanes; off < 64; off <<= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += __shfl_xor(acc[k], off, 64);
            cnt += __shfl_xor(cnt, off, 64);
        }
        const float den = cnt + 1e-12f;
        if (p == 0) {
            if (a.seq_mode[s] == FX_POOL_MEAN) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) acc[k] = acc[k] / den;
            }
            if (lane_on) fx_store<VEC>(a.out + b * a.out_ld + a.seq_out_off[s] + d0, acc);
            if (sub == 0) a.denom[b * a.n_seq + s] = den;
        }
    }
}

extern "C" int fx_emb_seq_pool_fwd(const float* table, int32_t D, const int32_t* ids,
                                   int64_t ids_ld, const int64_t* col_row_base,
                                   const int32_t* col_vocab, const int32_t* seq_col0,
                                   const int32_t* seq_len, const int32_t* seq_mode,
                                   const int64_t* seq_out_off, int32_t n_seq, float* out,
                                   int64_t out_ld, float* denom, int64_t B, fx_scalars* scal,
                                   int64_t table_ld, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_emb_seq_pool_fwd: D=%d not in [1,256]", D);
    if (table_ld <= 0) table_ld = D;
    FX_CHECK_ARG(n_seq >= 0 && B >= 0, "fx_emb_seq_pool_fwd: negative size");
    if (B == 0 || n_seq == 0) return FX_OK;
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(g.lanes <= 64, "fx_emb_seq_pool_fwd: D=%d needs %d lanes per row (max 64)", D,
                 g.lanes);
    FX_CHECK_ARG(table && ids && col_row_base && col_vocab && seq_col0 && seq_len && seq_mode &&
                     seq_out_off && out && denom && scal, "fx_emb_seq_pool_fwd: null pointer");
    FX_CHECK_ARG(out_ld % g.vec == 0, "fx_emb_seq_pool_fwd: out_ld=%lld not a multiple of %d",
                 (long long)out_ld, g.vec);
    FX_CHECK_ARG(table_ld >= D && table_ld % g.vec == 0,
                 "fx_emb_seq_pool_fwd: table_ld=%lld does not allow %d-wide row loads", (long long)table_ld, g.vec);
    SeqPoolArgs a{table, ids, ids_ld, col_row_base, col_vocab, seq_col0, seq_len, seq_mode,
                  seq_out_off, out, out_ld, denom, B, scal, D, n_seq, fx_log2i(g.lanes), table_ld};
    int64_t blocks = fx_ceil_div(B * (int64_t)n_seq, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (g.vec == 4) hipLaunchKernelGGL(k_emb_seq_pool_fwd<4>, grid, dim3(256), 0, s, a);
    else if (g.vec == 2) hipLaunchKernelGGL(k_emb_seq_pool_fwd<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_emb_seq_pool_fwd<1>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_numeric_grad: one 1024-thread block per numeric feature j; thread (grp, d) sums rows
// b = grp, grp + ngrp, ... then a fixed-order LDS reduction over the groups.
// ---------------------------------------------------------------------------------------------
#define FX_NUMGRAD_CHUNKS 16
__global__ __launch_bounds__(1024) void k_emb_numeric_grad(const float* dout, int64_t dout_ld,
                                                           const int64_t* num_out_off,
                                                           const float* dense, int64_t dense_ld,
                                                           int D, int Dp, int64_t B,
                                                           float* partial) {
    __shared__ float red[1024];
    const int j = blockIdx.x;
    const int d = threadIdx.x % Dp;
    const int grp = threadIdx.x / Dp;
    const int ngrp = 1024 / Dp;
    const int64_t off = num_out_off[j];
    const int64_t rows = (B + gridDim.y - 1) / gridDim.y;
    const int64_t b0 = (int64_t)blockIdx.y * rows;
    const int64_t b1 = (b0 + rows < B) ? b0 + rows : B;
    float acc = 0.f;
    if (d < D) {
        for (int64_t b = b0 + grp; b < b1; b += ngrp)
            acc = fmaf(dense[b * dense_ld + j], dout[b * dout_ld + off + d], acc);
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = ngrp >> 1; s > 0; s >>= 1) {
        if (grp < s) red[threadIdx.x] += red[threadIdx.x + s * Dp];
        __syncthreads();
    }
    if (grp == 0 && d < D)
        partial[((int64_t)blockIdx.y * gridDim.x + j) * D + d] = red[d];
}

__global__ __launch_bounds__(256) void k_emb_numeric_grad_final(const float* partial, int64_t n,
                                                                int chunks, float* out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += partial[(int64_t)c * n + i];
    out[i] = s;
}

extern "C" int fx_emb_numeric_grad(const float* dout, int64_t dout_ld,
                                   const int64_t* num_out_off, const float* dense,
                                   int64_t dense_ld, int32_t Fd, int32_t D, int64_t B,
                                   float* dnum_w, float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_emb_numeric_grad: D=%d not in [1,256]", D);
    if (Fd <= 0) return FX_OK;
    FX_CHECK_ARG(dout && num_out_off && dense && dnum_w, "fx_emb_numeric_grad: null pointer");
    int Dp = 1;
    while (Dp < D) Dp <<= 1;
    if (workspace) {   // two deterministic stages: FX_NUMGRAD_CHUNKS row chunks, then their sum
        hipLaunchKernelGGL(k_emb_numeric_grad, dim3(Fd, FX_NUMGRAD_CHUNKS), dim3(1024), 0,
                           fx_hip_stream(stream), dout, dout_ld, num_out_off, dense, dense_ld,
                           (int)D, Dp, B, workspace);
        const int64_t n = (int64_t)Fd * D;
        hipLaunchKernelGGL(k_emb_numeric_grad_final, dim3((unsigned)fx_ceil_div(n, 256)), dim3(256),
                           0, fx_hip_stream(stream), workspace, n, (int)FX_NUMGRAD_CHUNKS, dnum_w);
    } else {
        hipLaunchKernelGGL(k_emb_numeric_grad, dim3(Fd, 1), dim3(1024), 0, fx_hip_stream(stream),
                           dout, dout_ld, num_out_off, dense, dense_ld, (int)D, Dp, B, dnum_w);
    }
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// FM second-order term.  S lanes per sample (S = max(16, lanes-per-row)); S/lanes fields are
// walked in parallel; per-d field sums are completed with xor shuffles across the field-parallel
// lanes, then one more xor reduction over all S lanes gives 0.5*(sum_d s_d^2 - sum e^2).
// ---------------------------------------------------------------------------------------------
struct FmArgs {
    const float* emb;
    int64_t emb_ld;
    const float* addend;
    const float* g;
    float* out;
    float* demb;
    int64_t demb_ld;
    int64_t B;
    int32_t F, D, lanes_log2, S_log2, accumulate;
};

template <int VEC, bool BWD>
__global__ __launch_bounds__(256) void k_fm(FmArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int S = 1 << a.S_log2;
    const int ls = threadIdx.x & (S - 1);        // lane within sample group
    const int sub = ls & (lanes - 1);            // lane within row
    const int fpar = ls >> a.lanes_log2;         // field-parallel index
    const int nfpar = S >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int64_t spb = 256 >> a.S_log2;         // samples per block
    // every lane of a wave must take part in the shuffles: iterate on a wave-uniform bound
    const int64_t n_iter = (a.B + spb * gridDim.x - 1) / (spb * gridDim.x);
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t b = (it * gridDim.x + blockIdx.x) * spb + (threadIdx.x >> a.S_log2);
        const bool valid = b < a.B;
        float s[VEC], q = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] = 0.f;
        if (valid && lane_on) {
            const float* e = a.emb + b * a.emb_ld + d0;
            for (int f = fpar; f < a.F; f += nfpar) {
                float v[VEC];
                fx_load<VEC>(e + (int64_t)f * a.D, v);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s[k] += v[k];
                    q = fmaf(v[k], v[k], q);
                }
            }
        }
        // complete the per-d field sums across the field-parallel lanes
        for (int off = lanes; off < S; off <<= 1) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) s[k] += __shfl_xor(s[k], off, 64);
        }
        if constexpr (!BWD) {
            float t = -q;
            if (fpar == 0) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) t = fmaf(s[k], s[k], t);
            }
            for (int off = 1; off < S; off <<= 1) t += __shfl_xor(t, off, 64);
            if (valid && ls == 0) {
                float r = 0.5f * t;
                if (a.addend) r += a.addend[b];
                a.out[b] = r;
            }
        } else {
            if (valid && lane_on) {
                const float gb = a.g[b];
                const float* e = a.emb + b * a.emb_ld + d0;
                float* de = a.demb + b * a.demb_ld + d0;
                for (int f = fpar; f < a.F; f += nfpar) {
                    float v[VEC], o[VEC];
                    fx_load<VEC>(e + (int64_t)f * a.D, v);
                    if (a.accumulate) fx_load<VEC>(de + (int64_t)f * a.D, o);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        const float t = gb * (s[k] - v[k]);
                        o[k] = a.accumulate ? o[k] + t : t;
                    }
                    fx_store<VEC>(de + (int64_t)f * a.D, o);
                }
            }
        }
    }
}

static int fx_fm_launch(bool bwd, FmArgs a, fx_stream_t stream) {
    const FxRowGeom g = fx_row_geom(a.D);
    a.lanes_log2 = fx_log2i(g.lanes);
    int S = g.lanes < 16 ? 16 : g.lanes;
    a.S_log2 = fx_log2i(S);
    int64_t blocks = fx_ceil_div(a.B, 256 / S);
    if (blocks > 256 * 16) blocks = 256 * 16;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (!bwd) {
        if (g.vec == 4) hipLaunchKernelGGL((k_fm<4, false>), grid, dim3(256), 0, s, a);
        else if (g.vec == 2) hipLaunchKernelGGL((k_fm<2, false>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_fm<1, false>), grid, dim3(256), 0, s, a);
    } else {
        if (g.vec == 4) hipLaunchKernelGGL((k_fm<4, true>), grid, dim3(256), 0, s, a);
        else if (g.vec == 2) hipLaunchKernelGGL((k_fm<2, true>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((k_fm<1, true>), grid, dim3(256), 0, s, a);
    }
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_fm_fwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D,
                         const float* addend, float* out, int64_t B, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256 && F >= 1, "fx_fm_fwd: bad F=%d / D=%d", F, D);
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(emb && out, "fx_fm_fwd: null pointer");
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(emb_ld % g.vec == 0, "fx_fm_fwd: emb_ld not a multiple of %d", g.vec);
    FmArgs a{emb, emb_ld, addend, nullptr, out, nullptr, 0, B, F, D, 0, 0, 0};
    return fx_fm_launch(false, a, stream);
}

extern "C" int fx_fm_bwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D, const float* g,
                         float* demb, int64_t demb_ld, int32_t accumulate, int64_t B,
                         fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256 && F >= 1, "fx_fm_bwd: bad F=%d / D=%d", F, D);
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(emb && g && demb, "fx_fm_bwd: null pointer");
    const FxRowGeom geo = fx_row_geom(D);
    FX_CHECK_ARG(emb_ld % geo.vec == 0 && demb_ld % geo.vec == 0,
                 "fx_fm_bwd: leading dimensions not a multiple of %d", geo.vec);
    FmArgs a{emb, emb_ld, nullptr, g, nullptr, demb, demb_ld, B, F, D, 0, 0, accumulate};
    return fx_fm_launch(true, a, stream);
}

// ---------------------------------------------------------------------------------------------
// LR first-order term: 16 lanes per sample walk the C id columns (4-byte rows of the D=1 table)
// and the Fd numeric columns, then an xor reduction.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_lr_fwd(const float* table1, const int32_t* ids,
                                                int64_t ids_ld, const int64_t* col_row_base,
                                                const int32_t* col_vocab, int C,
                                                const float* dense, int64_t dense_ld,
                                                const float* num_w1, int Fd, const float* bias,
                                                float* out, int64_t B, fx_scalars* scal,
                                                int64_t table1_ld) {
    const int ls = threadIdx.x & 15;
    const int64_t n_iter = (B + 16 * (int64_t)gridDim.x - 1) / (16 * (int64_t)gridDim.x);
    for (int64_t it = 0; it < n_iter; ++it) {
        const int64_t b = (it * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 4);
        const bool valid = b < B;
        float acc = 0.f;
        if (valid) {
            for (int c = ls; c < C; c += 16) {
                const int32_t id = ids[b * ids_ld + c];
                if (id >= 0 && id < col_vocab[c]) acc += table1[(col_row_base[c] + id) * table1_ld];
                else atomicOr(&scal->err_flag, FX_FLAG_BAD_ID);
            }
            for (int j = ls; j < Fd; j += 16) acc = fmaf(dense[b * dense_ld + j], num_w1[j], acc);
        }
        for (int off = 1; off < 16; off <<= 1) acc += __shfl_xor(acc, off, 64);
        if (valid && ls == 0) out[b] = acc + (bias ? bias[0] : 0.f);
    }
}

extern "C" int fx_lr_fwd(const float* table1, const int32_t* ids, int64_t ids_ld,
                         const int64_t* col_row_base, const int32_t* col_vocab, int32_t C,
                         const float* dense, int64_t dense_ld, const float* num_w1, int32_t Fd,
                         const float* bias, float* out, int64_t B, fx_scalars* scal,
                         int64_t table1_ld, fx_stream_t stream) {
    FX_CHECK_ARG(C >= 0 && Fd >= 0, "fx_lr_fwd: negative size");
    if (table1_ld <= 0) table1_ld = 1;
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(out && scal, "fx_lr_fwd: null out/scal");
    FX_CHECK_ARG(C == 0 || (table1 && ids && col_row_base && col_vocab),
                 "fx_lr_fwd: null sparse argument");
    FX_CHECK_ARG(Fd == 0 || (dense && num_w1), "fx_lr_fwd: null numeric argument");
    int64_t blocks = fx_ceil_div(B, 16);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_lr_fwd, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream),
                       table1, ids, ids_ld, col_row_base, col_vocab, (int)C, dense, dense_ld,
                       num_w1, (int)Fd, bias, out, B, scal, table1_ld);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// Pairwise dot interaction (DLRM "dot"): out[b, p(i,j)] = <e[b,i,:], e[b,j,:]> for i < j, pairs in
// row-major upper-triangle order — InnerProductInteraction "inner_product",
// fuxictr/pytorch/layers/interactions/inner_product.py:63-66 (bmm + triu masked_select).
// One workgroup stages a sample's F x D embeddings in LDS; backward:
// de[b,i,:] = sum_{j != i} g[b, p(min,max)] e[b,j,:].
// ---------------------------------------------------------------------------------------------
#define FX_DOT_MAX_FD 4096

__device__ __forceinline__ int fx_pair_index(int i, int j, int F) {  // i < j
    return i * F - (i * (i + 1)) / 2 + (j - i - 1);
}

// tail > 0 (DLRM: the dense vector is the LAST field and is also concatenated to the products,
// DLRM.py:117-120): out[b, P .. P + D) = e[b, F-1, :] and out[b, P + D .. P + D + tail - D) = 0 ride along, so
// the row [products | dense | zero padding] leaves this launch ready for the top tower (out_ld columns).
__global__ __launch_bounds__(256) void k_dot_interact_fwd(const float* emb, int64_t emb_ld, int F,
                                                          int D, int64_t B, float* out, int64_t out_ld,
                                                          int tail) {
    __shared__ float e[FX_DOT_MAX_FD];
    const int P = F * (F - 1) / 2;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        for (int t = threadIdx.x; t < F * D; t += 256) e[t] = emb[b * emb_ld + t];
        __syncthreads();
        for (int t = threadIdx.x; t < tail; t += 256) out[b * out_ld + P + t] = t < D ? e[(F - 1) * D + t] : 0.f;
        for (int p = threadIdx.x; p < P; p += 256) {
            // invert p -> (i, j): rows of the upper triangle have F-1, F-2, ... entries
            int i = 0, rem = p;
            while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
            const int j = i + 1 + rem;
            float acc = 0.f;
            for (int d = 0; d < D; ++d) acc = fmaf(e[i * D + d], e[j * D + d], acc);
            out[b * out_ld + p] = acc;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_dot_interact_bwd(const float* emb, int64_t emb_ld,
                                                          const float* g, int64_t g_ld, int tail, int F,
                                                          int D, int64_t B, float* demb,
                                                          int64_t demb_ld) {
    __shared__ float e[FX_DOT_MAX_FD];
    __shared__ float gs[FX_DOT_MAX_FD];
    const int P = F * (F - 1) / 2;
    for (int64_t b = blockIdx.x; b < B; b += gridDim.x) {
        for (int t = threadIdx.x; t < F * D; t += 256) e[t] = emb[b * emb_ld + t];
        for (int t = threadIdx.x; t < P; t += 256) gs[t] = g[b * g_ld + t];
        __syncthreads();
        for (int t = threadIdx.x; t < F * D; t += 256) {
            const int i = t / D, d = t - i * D;
            float acc = 0.f;
            for (int j = 0; j < F; ++j) {
                if (j == i) continue;
                const int p = j > i ? fx_pair_index(i, j, F) : fx_pair_index(j, i, F);
                acc = fmaf(gs[p], e[j * D + d], acc);
            }
            if (tail && i == F - 1) acc += g[b * g_ld + P + d];      // the concatenated copy's gradient
            demb[b * demb_ld + t] = acc;
        }
        __syncthreads();
    }
}

// The same two kernels on the matrix cores for F <= 32 fields of even D <= 32 (DLRM: 27 x 16): per
// sample the work is a 27 x 27 x 16 product — far too little for a workgroup with two barriers per
// sample (the kernels above: 17.7 / 46.2 us of the 0.93 ms DLRM step).  Here ONE WAVE owns a sample:
// E goes to a wave-private LDS tile, E E^T is D/2 MFMAs (v_mfma_f32_32x32x2_f32, A and B fragments
// are the same LDS read), the backward (G + G^T) E is 16 MFMAs on the symmetrised gradient tile.
typedef float fx_dot_f32x16 __attribute__((ext_vector_type(16)));
#define FX_DOT_LD 33

__global__ __launch_bounds__(256) void k_dot_interact_fwd_mfma(const float* emb, int64_t emb_ld,
                                                               int F, int D, int64_t B, float* out,
                                                               int64_t out_ld, int tail) {
    __shared__ float Es_[4][32 * FX_DOT_LD];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Es = Es_[wave];
    for (int t = lane; t < 32 * FX_DOT_LD; t += 64) Es[t] = 0.f;
    const int P = F * (F - 1) / 2;
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
        for (int t = lane; t < F * D; t += 64) {
            const int i = t / D;
            Es[i * FX_DOT_LD + (t - i * D)] = emb[b * emb_ld + t];
        }
        fx_dot_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kk = 0; kk < (D >> 1); ++kk) {
            const float a = Es[l31 * FX_DOT_LD + 2 * kk + half];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc, 0, 0, 0);
        }
        const int j = l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < j && j < F) out[b * out_ld + fx_pair_index(i, j, F)] = acc[r];
        }
        for (int t = lane; t < tail; t += 64)
            out[b * out_ld + P + t] = t < D ? Es[(F - 1) * FX_DOT_LD + t] : 0.f;
    }
}

__global__ __launch_bounds__(256) void k_dot_interact_bwd_mfma(const float* emb, int64_t emb_ld,
                                                               const float* g, int64_t g_ld, int tail,
                                                               int F, int D, int64_t B, float* demb,
                                                               int64_t demb_ld) {
    __shared__ float Es_[4][32 * FX_DOT_LD];
    __shared__ float Gs_[4][32 * FX_DOT_LD];
    __shared__ unsigned char pi_[512], pj_[512];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Es = Es_[wave];
    float* Gs = Gs_[wave];
    const int P = F * (F - 1) / 2;
    for (int t = lane; t < 32 * FX_DOT_LD; t += 64) {
        Es[t] = 0.f;
        Gs[t] = 0.f;
    }
    for (int p = threadIdx.x; p < P; p += 256) {       // p -> (i, j), i < j, row-major upper triangle
        int i = 0, rem = p;
        while (rem >= F - 1 - i) { rem -= F - 1 - i; ++i; }
        pi_[p] = (unsigned char)i;
        pj_[p] = (unsigned char)(i + 1 + rem);
    }
    __syncthreads();
    for (int64_t b = (int64_t)blockIdx.x * 4 + wave; b < B; b += (int64_t)gridDim.x * 4) {
        for (int t = lane; t < F * D; t += 64) {
            const int i = t / D;
            Es[i * FX_DOT_LD + (t - i * D)] = emb[b * emb_ld + t];
        }
        for (int p = lane; p < P; p += 64) {
            const float v = g[b * g_ld + p];
            const int i = pi_[p], j = pj_[p];
            Gs[i * FX_DOT_LD + j] = v;
            Gs[j * FX_DOT_LD + i] = v;
        }
        fx_dot_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int nk = (F + 1) >> 1;
        for (int kk = 0; kk < nk; ++kk) {
            const float a = Gs[l31 * FX_DOT_LD + 2 * kk + half];
            const float e = Es[(2 * kk + half) * FX_DOT_LD + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, e, acc, 0, 0, 0);
        }
        const int d = l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < F && d < D)
                demb[b * demb_ld + i * D + d] = (tail && i == F - 1) ? acc[r] + g[b * g_ld + P + d] : acc[r];
        }
    }
}

static bool fx_dot_mfma_ok(int F, int D) {
    static const bool on = []() {   // FX_DOT_MFMA=0: the workgroup-per-sample kernels (A/B runs)
        const char* e = getenv("FX_DOT_MFMA");
        return !(e && atoi(e) == 0);
    }();
    return on && F <= 32 && D <= 32 && D % 2 == 0;
}

extern "C" int fx_dot_interact_fwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D,
                                   int64_t B, float* out, int64_t out_ld, int32_t tail,
                                   fx_stream_t stream) {
    FX_CHECK_ARG(F >= 2 && D >= 1 && F * D <= FX_DOT_MAX_FD && F * (F - 1) / 2 <= FX_DOT_MAX_FD,
                 "fx_dot_interact_fwd: F=%d D=%d outside the supported range", F, D);
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(emb && out, "fx_dot_interact_fwd: null pointer");
    FX_CHECK_ARG(tail == 0 || tail >= D, "fx_dot_interact_fwd: tail must be 0 or >= D");
    FX_CHECK_ARG(out_ld >= (int64_t)F * (F - 1) / 2 + tail, "fx_dot_interact_fwd: out_ld too small");
    if (fx_dot_mfma_ok(F, D)) {
        int64_t wgs = fx_ceil_div(B, 4);
        if (wgs > 2048) wgs = 2048;
        hipLaunchKernelGGL(k_dot_interact_fwd_mfma, dim3((unsigned)wgs), dim3(256), 0,
                           fx_hip_stream(stream), emb, emb_ld, (int)F, (int)D, B, out, out_ld, (int)tail);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    int64_t blocks = B < 8192 ? B : 8192;
    hipLaunchKernelGGL(k_dot_interact_fwd, dim3((unsigned)blocks), dim3(256), 0,
                       fx_hip_stream(stream), emb, emb_ld, (int)F, (int)D, B, out, out_ld, (int)tail);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dot_interact_bwd(const float* emb, int64_t emb_ld, const float* g, int64_t g_ld,
                                   int32_t tail, int32_t F, int32_t D, int64_t B, float* demb,
                                   int64_t demb_ld, fx_stream_t stream) {
    FX_CHECK_ARG(F >= 2 && D >= 1 && F * D <= FX_DOT_MAX_FD && F * (F - 1) / 2 <= FX_DOT_MAX_FD,
                 "fx_dot_interact_bwd: F=%d D=%d outside the supported range", F, D);
    if (B <= 0) return FX_OK;
    FX_CHECK_ARG(emb && g && demb, "fx_dot_interact_bwd: null pointer");
    if (fx_dot_mfma_ok(F, D)) {
        int64_t wgs = fx_ceil_div(B, 4);
        if (wgs > 2048) wgs = 2048;
        hipLaunchKernelGGL(k_dot_interact_bwd_mfma, dim3((unsigned)wgs), dim3(256), 0,
                           fx_hip_stream(stream), emb, emb_ld, g, g_ld, (int)tail, (int)F, (int)D, B, demb,
                           demb_ld);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    int64_t blocks = B < 8192 ? B : 8192;
    hipLaunchKernelGGL(k_dot_interact_bwd, dim3((unsigned)blocks), dim3(256), 0,
                       fx_hip_stream(stream), emb, emb_ld, g, g_ld, (int)tail, (int)F, (int)D, B, demb,
                       demb_ld);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
