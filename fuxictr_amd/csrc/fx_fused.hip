// fx_fused.hip — the sparse front end and back end of the training step as a handful of launches
// (BASELINE.json north_star: "the FeatureEmbedding multi-field sparse lookup ... fused straight into
// the feature-interaction layers").  Replaces, for one FeatureEmbeddingDict whose id columns own
// disjoint tables (every categorical schema of the BASELINE configs):
//
//   forward   fx_dedup_catchup   sort + (fused) begin-step  |  unique rows + exact-mode Adam catch-up of
//                                EVERY table group that shares the id plan (the D=16 tables and the
//                                D=1 tables of LogisticRegression) — 2 launches (were 6)
//             fx_emb_fm_fwd      gather + numeric expansion + first-order term + FM second-order term,
//                                one wave per sample, the record is written once — 1 launch (were 3)
//   backward  fx_emb_fm_bwd      FM backward folded into the run-reduce of the record's gradient, the
//                                D=1 rows reduced in the same pass, ||G||^2 partials fused, work
//                                balanced over the sorted lookups (no short/long split); numeric
//                                weights / LR bias in a second launch — 2 launches (were 13)
//   update    fx_sparse_adam_multi   the row update of all those table groups — 1 launch (were 2)
//   inputs    fx_pack_columns_multi  ids / numerics / label casts of a batch — 1 launch (were 3)
//
// Reference lines replaced (paths relative to the reference checkout):
//   fuxictr/pytorch/layers/embeddings/feature_embedding.py:261-297, :230-259 (lookup loop, stack)
//   fuxictr/pytorch/layers/blocks/logistic_regression.py:46-59 (second D=1 embedding pass, sum, bias)
//   fuxictr/pytorch/layers/interactions/inner_product.py:55-62 (product_sum) and their autograd,
//   aten::embedding_dense_backward at rank_model.py:320, the table part of clip_grad_norm_
//   (rank_model.py:321) and of torch.optim.Adam.step (rank_model.py:322).
#include "fx_common.h"

#include <rocprim/rocprim.hpp>

// ---------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------
#define FX_REPLAY_MAX2 256   // same truncation as k_adam_catchup (fx_sparse.hip)

struct FxTableDev {
    float* table;
    float* m;
    float* v;
    int32_t* last_step;
    const float* G;        // update kernels only
    int32_t D, vec, lanes_log2;
};

#define FX_MAX_TABLES 4

// zero-gradient Adam replay of one row (see k_adam_catchup): lanes sub < lanes of the group
template <int VEC>
__device__ __forceinline__ void fx_catchup_row(const FxTableDev& t, int64_t row, int sub,
                                               const fx_scalars& sc, int upto, double lb1,
                                               double lb2) {
    const int lanes = 1 << t.lanes_log2;
    if (sub >= lanes) return;
    const int last = t.last_step[row];
    const int k_steps = upto - last;
    if (k_steps <= 0) return;
    const int d0 = sub * VEC;
    if (d0 < t.D) {
        float p[VEC], m[VEC], v[VEC];
        const int64_t o = row * t.D + d0;
        fx_load<VEC>(t.m + o, m);
        fx_load<VEC>(t.v + o, v);
        bool any = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) any = any || (m[k] != 0.f) || (v[k] != 0.f);
        if (any) {
            const float w1 = 1.f - sc.beta1;
            fx_load<VEC>(t.table + o, p);
            const int kk = k_steps < FX_REPLAY_MAX2 ? k_steps : FX_REPLAY_MAX2;
            float pw1 = (float)exp2(lb1 * (double)last);
            float pw2 = (float)exp2(lb2 * (double)last);
            for (int j = 0; j < kk; ++j) {
                pw1 *= sc.beta1;
                pw2 *= sc.beta2;
                const float step_size = sc.lr / (1.f - pw1);
                const float bc2s = sqrtf(1.f - pw2);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    m[k] = m[k] + w1 * (0.f - m[k]);
                    v[k] = v[k] * sc.beta2;
                    p[k] = p[k] - step_size * (m[k] / (sqrtf(v[k]) / bc2s + sc.eps));
                }
            }
            if (k_steps > kk) {
                const float f1 = (float)exp2(lb1 * (double)(k_steps - kk));
                const float f2 = (float)exp2(lb2 * (double)(k_steps - kk));
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    m[k] *= f1;
                    v[k] *= f2;
                }
            }
            fx_store<VEC>(t.table + o, p);
            fx_store<VEC>(t.m + o, m);
            fx_store<VEC>(t.v + o, v);
        }
    }
    if (sub == 0) t.last_step[row] = upto;
}

// ---------------------------------------------------------------------------------------------
// fx_dedup_catchup, launch 1: one workgroup sorts one id column in LDS (as k_sort_columns of
// fx_sparse.hip); block 0 also opens the optimizer step (fx_opt_begin_step fused).
// ---------------------------------------------------------------------------------------------
template <int IPT>
__global__ __launch_bounds__(1024) void k_sort_columns2(const int32_t* ids, int64_t ids_ld, int64_t B,
                                                        const int64_t* col_row_base,
                                                        const int32_t* col_vocab,
                                                        const int32_t* col_pad, int C,
                                                        uint32_t* sorted_key, uint32_t* sorted_pos,
                                                        uint32_t* col_scan, uint32_t* col_cnt,
                                                        fx_scalars* begin_scal) {
    using Sort = rocprim::block_radix_sort<uint32_t, 1024, IPT, uint32_t>;
    using Scan = rocprim::block_scan<uint32_t, 1024>;
    __shared__ typename Sort::storage_type storage;
    __shared__ typename Scan::storage_type scan_storage;
    __shared__ uint32_t lastk[1024];
    const int c = blockIdx.x;
    if (begin_scal != nullptr && c == 0 && threadIdx.x == 0) {
        // torch.optim.Adam: bias_correction1 = 1 - beta1 ** step (python double), step_size =
        // lr / bias_correction1, bias_correction2_sqrt = (1 - beta2 ** step) ** 0.5
        fx_scalars* sc = begin_scal;
        const int t = sc->step + 1;
        sc->step = t;
        const double b1 = (double)sc->beta1, b2 = (double)sc->beta2;
        const double bc1 = 1.0 - pow(b1, (double)t);
        const double bc2 = 1.0 - pow(b2, (double)t);
        sc->bc1 = (float)bc1;
        sc->bc2_sqrt = (float)sqrt(bc2);
        sc->step_size = (float)((double)sc->lr / bc1);
    }
    const int32_t V = col_vocab[c], pad = col_pad[c];
    int bits = 1;
    while ((1u << bits) < (uint32_t)V && bits < 31) ++bits;
    const uint32_t fill = (1u << bits) - 1u;   // >= every real id; ties keep real items first
    uint32_t k[IPT], v[IPT];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t b = (int64_t)threadIdx.x * IPT + j;
        k[j] = fill;
        v[j] = 0xFFFFFFFFu;
        if (b < B) {
            const int32_t id = ids[b * ids_ld + c];
            const bool in_range = id >= 0 && id < V;
            k[j] = in_range ? (uint32_t)id : 0u;
            if (in_range && id != pad) v[j] = (uint32_t)(b * C + c);
        }
    }
    Sort().sort(k, v, storage, 0, bits);
    lastk[threadIdx.x] = k[IPT - 1];
    __syncthreads();
    uint32_t flag[IPT], h = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = (int64_t)threadIdx.x * IPT + j;
        const uint32_t prev = j > 0 ? k[j - 1] : (threadIdx.x > 0 ? lastk[threadIdx.x - 1] : 0u);
        flag[j] = (i < B && (i == 0 || k[j] != prev)) ? 1u : 0u;
        h += flag[j];
    }
    uint32_t before = 0, total = 0;
    Scan().exclusive_scan(h, before, 0u, total, scan_storage);
    if (threadIdx.x == 0) col_cnt[c] = total;
    const uint32_t base = (uint32_t)col_row_base[c];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = (int64_t)threadIdx.x * IPT + j;
        before += flag[j];
        if (i < B) {
            sorted_key[(int64_t)c * B + i] = base + k[j];
            sorted_pos[(int64_t)c * B + i] = v[j];
            col_scan[(int64_t)c * B + i] = before;
        }
    }
}

// launch 2: one lane group per sorted lookup.  The group of a run's FIRST lookup owns the unique
// row: it writes uniq_row / seg_start and replays the row's missed zero-gradient Adam steps in
// every table group that shares the id plan, so the gather that follows reads current rows.
struct FinishArgs {
    const uint32_t* key;
    const uint32_t* col_scan;
    const uint32_t* col_cnt;
    uint32_t* uniq_row;
    uint32_t* seg_start;
    int32_t* n_unique;
    uint32_t* sorted_uid;
    FxTableDev t[FX_MAX_TABLES];
    const fx_scalars* scal;
    int64_t B;
    int32_t C, n_tables, group_log2, upto_offset;
};

__global__ __launch_bounds__(256) void k_finish_catchup(FinishArgs a) {
    __shared__ uint32_t off[257];
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int c = 0; c < a.C; ++c) {
            off[c] = acc;
            acc += a.col_cnt[c];
        }
        off[a.C] = acc;
    }
    __syncthreads();
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t ipb = 256 >> a.group_log2;
    const int64_t n = a.B * a.C;
    fx_scalars sc;
    int upto = 0;
    double lb1 = 0.0, lb2 = 0.0;
    if (a.n_tables > 0) {
        sc = *a.scal;
        upto = sc.step + a.upto_offset;
        lb1 = log2((double)sc.beta1);
        lb2 = log2((double)sc.beta2);
    }
    for (int64_t i = (int64_t)blockIdx.x * ipb + (threadIdx.x >> a.group_log2); i < n;
         i += (int64_t)gridDim.x * ipb) {
        const int c = (int)(i / a.B);
        const uint32_t k = a.key[i];
        const uint32_t u = off[c] + a.col_scan[i];
        const bool head = (i == 0) || (a.key[i - 1] != k);
        if (sub == 0) {
            if (a.sorted_uid) a.sorted_uid[i] = u - 1;
            if (head) {
                a.uniq_row[u - 1] = k;
                a.seg_start[u - 1] = (uint32_t)i;
            }
            if (i == n - 1) {
                a.seg_start[u] = (uint32_t)(i + 1);
                *a.n_unique = (int32_t)u;
            }
        }
        if (!head) continue;
        for (int t = 0; t < a.n_tables; ++t) {
            const FxTableDev& tb = a.t[t];
            if (tb.vec == 4) fx_catchup_row<4>(tb, (int64_t)k, sub, sc, upto, lb1, lb2);
            else if (tb.vec == 2) fx_catchup_row<2>(tb, (int64_t)k, sub, sc, upto, lb1, lb2);
            else fx_catchup_row<1>(tb, (int64_t)k, sub, sc, upto, lb1, lb2);
        }
    }
}

static int fx_fill_tables(const fx_row_state* tables_host, int32_t n_tables, FxTableDev* out,
                          int* group_log2, const char* who, bool need_state) {
    int gl = 0;
    for (int t = 0; t < n_tables; ++t) {
        const fx_row_state& h = tables_host[t];
        if (h.D < 1 || h.D > 256) {
            fx_set_error("%s: table %d has D=%d outside [1,256]", who, t, h.D);
            return FX_ERR_INVALID;
        }
        if (!h.table || (need_state && (!h.m || !h.v || !h.last_step))) {
            fx_set_error("%s: table %d has a null pointer", who, t);
            return FX_ERR_INVALID;
        }
        const FxRowGeom g = fx_row_geom(h.D);
        int ll = 0;
        while ((1 << ll) < g.lanes) ++ll;
        out[t].table = h.table;
        out[t].m = h.m;
        out[t].v = h.v;
        out[t].last_step = h.last_step;
        out[t].G = h.G;
        out[t].D = h.D;
        out[t].vec = g.vec;
        out[t].lanes_log2 = ll;
        if (ll > gl) gl = ll;
    }
    if (gl > 6) {
        fx_set_error("%s: rows of more than 64 lanes are not supported here", who);
        return FX_ERR_UNSUPPORTED;
    }
    *group_log2 = gl;
    return FX_OK;
}

extern "C" int fx_dedup_catchup(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                                const int64_t* col_row_base, const int32_t* col_vocab,
                                const int32_t* col_pad, void* workspace, size_t workspace_bytes,
                                uint32_t* sorted_key, uint32_t* sorted_pos, uint32_t* uniq_row,
                                uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                                fx_scalars* begin_scal, const fx_row_state* tables_host,
                                int32_t n_tables, int32_t upto_offset, const fx_scalars* scal,
                                fx_stream_t stream) {
    FX_CHECK_ARG(B >= 1 && B <= 8192 && C >= 1 && C <= 256,
                 "fx_dedup_catchup: B=%lld (1..8192) / C=%d (1..256) outside the column fast path",
                 (long long)B, C);
    FX_CHECK_ARG(n_tables >= 0 && n_tables <= FX_MAX_TABLES, "fx_dedup_catchup: n_tables=%d > %d",
                 n_tables, FX_MAX_TABLES);
    FX_CHECK_ARG(ids && col_row_base && col_vocab && col_pad && workspace && sorted_key &&
                     sorted_pos && uniq_row && seg_start && n_unique,
                 "fx_dedup_catchup: null pointer");
    FX_CHECK_ARG(n_tables == 0 || (tables_host && scal), "fx_dedup_catchup: tables without scal");
    const int64_t n = B * (int64_t)C;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) / 256 * 256;
    FX_CHECK_ARG(workspace_bytes >= 2 * arr, "fx_dedup_catchup: workspace too small (%zu < %zu)",
                 workspace_bytes, 2 * arr);
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* col_cnt = reinterpret_cast<uint32_t*>(w);
    uint32_t* col_scan = reinterpret_cast<uint32_t*>(w + arr);
    FinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, fa.t, &gl, "fx_dedup_catchup", true);
    if (st != FX_OK) return st;
    hipStream_t s = fx_hip_stream(stream);
#define FX_SORT2(IPT)                                                                             \
    hipLaunchKernelGGL(k_sort_columns2<IPT>, dim3(C), dim3(1024), 0, s, ids, ids_ld, B,           \
                       col_row_base, col_vocab, col_pad, (int)C, sorted_key, sorted_pos, col_scan, \
                       col_cnt, begin_scal)
    if (B <= 1024) FX_SORT2(1);
    else if (B <= 2048) FX_SORT2(2);
    else if (B <= 4096) FX_SORT2(4);
    else FX_SORT2(8);
#undef FX_SORT2
    FX_CHECK_LAUNCH();
    fa.key = sorted_key;
    fa.col_scan = col_scan;
    fa.col_cnt = col_cnt;
    fa.uniq_row = uniq_row;
    fa.seg_start = seg_start;
    fa.n_unique = n_unique;
    fa.sorted_uid = sorted_uid;
    fa.scal = scal;
    fa.B = B;
    fa.C = C;
    fa.n_tables = n_tables;
    fa.group_log2 = gl;
    fa.upto_offset = upto_offset;
    int64_t blocks = fx_ceil_div(n, 256 >> gl);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_finish_catchup, dim3((unsigned)blocks), dim3(256), 0, s, fa);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_fm_fwd: one wave per sample.  The wave's 64 lanes are 64/lanes lane groups; group g walks
// the record's items g, g+G, ... (an id lookup or a numeric expansion), writes the item's row into
// its slot of the record and keeps the per-dimension field sums and the sum of squares in
// registers; the first-order term is summed by lane l over id column l / numeric column l.  Two
// xor butterflies finish the sample: nothing but the record, the field sums S (for the backward) and
// three scalars per sample is written.
// ---------------------------------------------------------------------------------------------
struct EmbFmArgs {
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
    const float* table1;
    const float* num_w1;
    const float* bias1;
    float* lr_out;
    float* fm_out;
    float* fm_lr_out;
    float* S;
    fx_scalars* scal;
    int32_t D, C, Fd, lanes_log2;
};

template <int VEC>
__global__ __launch_bounds__(256) void k_emb_fm_fwd(EmbFmArgs a) {
    const int lane = threadIdx.x & 63;
    const int lanes = 1 << a.lanes_log2;
    const int sub = lane & (lanes - 1);
    const int grp = lane >> a.lanes_log2;
    const int ngrp = 64 >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int R = a.C + a.Fd;
    const bool want_fm = a.fm_out != nullptr || a.fm_lr_out != nullptr || a.S != nullptr;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < a.B;
         b += (int64_t)gridDim.x * 4) {                                  // wave-uniform
        float s[VEC], q = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] = 0.f;
        for (int r = grp; r < R; r += ngrp) {
            float val[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) val[k] = 0.f;
            int64_t off;
            if (r < a.C) {
                const int32_t id = a.ids[b * a.ids_ld + r];
                off = a.col_out_off[r];
                if (id >= 0 && id < a.col_vocab[r]) {
                    if (lane_on) fx_load<VEC>(a.table + (a.col_row_base[r] + id) * a.D + d0, val);
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
            if (lane_on) {
                fx_store<VEC>(a.out + b * a.out_ld + off + d0, val);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s[k] += val[k];
                    q = fmaf(val[k], val[k], q);
                }
            }
        }
        float lr = 0.f;
        if (a.lr_out != nullptr || a.fm_lr_out != nullptr) {
            for (int c = lane; c < a.C; c += 64) {
                const int32_t id = a.ids[b * a.ids_ld + c];
                if (id >= 0 && id < a.col_vocab[c]) lr += a.table1[a.col_row_base[c] + id];
            }
            for (int j = lane; j < a.Fd; j += 64)
                lr = fmaf(a.dense[b * a.dense_ld + j], a.num_w1[j], lr);
            lr = fx_wave_sum(lr);
            if (a.bias1) lr += a.bias1[0];
        }
        float fm = 0.f;
        if (want_fm) {
            // per-dimension field sums: combine the lane groups (fixed butterfly order)
            for (int o = lanes; o < 64; o <<= 1) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s[k] += __shfl_xor(s[k], o, 64);
            }
            float t = -q;
            if (grp == 0 && lane_on) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) t = fmaf(s[k], s[k], t);
            }
            fm = 0.5f * fx_wave_sum(t);
            if (a.S && grp == 0 && lane_on) fx_store<VEC>(a.S + b * a.D + d0, s);
        }
        if (lane == 0) {
            if (a.lr_out) a.lr_out[b] = lr;
            if (a.fm_out) a.fm_out[b] = fm;
            if (a.fm_lr_out) a.fm_lr_out[b] = fm + lr;
        }
    }
}

extern "C" int fx_emb_fm_fwd(const float* table, int32_t D, const int32_t* ids, int64_t ids_ld,
                             const int64_t* col_row_base, const int32_t* col_vocab,
                             const int64_t* col_out_off, int32_t C, const float* dense,
                             int64_t dense_ld, const float* num_w, const int64_t* num_out_off,
                             int32_t Fd, float* out, int64_t out_ld, int64_t B,
                             const float* table1, const float* num_w1, const float* bias1,
                             float* lr_out, float* fm_out, float* fm_lr_out, float* S,
                             fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_emb_fm_fwd: D=%d not in [1,256]", D);
    FX_CHECK_ARG(C >= 0 && Fd >= 0 && B >= 0, "fx_emb_fm_fwd: negative size");
    if (B == 0 || C + Fd == 0) return FX_OK;
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(g.lanes <= 64, "fx_emb_fm_fwd: D=%d needs %d lanes per row (max 64)", D, g.lanes);
    FX_CHECK_ARG(out && scal, "fx_emb_fm_fwd: null out/scal");
    FX_CHECK_ARG(C == 0 || (table && ids && col_row_base && col_vocab && col_out_off),
                 "fx_emb_fm_fwd: null sparse argument");
    FX_CHECK_ARG(Fd == 0 || (dense && num_w && num_out_off), "fx_emb_fm_fwd: null numeric argument");
    const bool want_lr = lr_out != nullptr || fm_lr_out != nullptr;
    FX_CHECK_ARG(!want_lr || ((C == 0 || table1) && (Fd == 0 || num_w1)),
                 "fx_emb_fm_fwd: first-order term requested without its D=1 table / numeric weights");
    FX_CHECK_ARG(out_ld % g.vec == 0, "fx_emb_fm_fwd: out_ld=%lld not a multiple of %d",
                 (long long)out_ld, g.vec);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    EmbFmArgs a{table, ids, ids_ld, col_row_base, col_vocab, col_out_off, dense, dense_ld, num_w,
                num_out_off, out, out_ld, B, table1, num_w1, bias1, lr_out, fm_out, fm_lr_out, S,
                scal, D, C, Fd, ll};
    int64_t blocks = fx_ceil_div(B, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (g.vec == 4) hipLaunchKernelGGL(k_emb_fm_fwd<4>, grid, dim3(256), 0, s, a);
    else if (g.vec == 2) hipLaunchKernelGGL(k_emb_fm_fwd<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_emb_fm_fwd<1>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_fm_bwd, launch 1: gradient of every unique row.
// Workgroup w owns the unique rows [w*RPB, (w+1)*RPB) and therefore ONE contiguous range of the
// sorted lookups; that range is cut into NG = 256/lanes equal pieces, lane group g sums the lookups
// of piece g in ascending order.  A row that lies inside one piece is finished by its group; a row
// spread over several pieces (a hot row of a tiny table: ~1365 lookups at B = 4096) is combined
// from the groups' partial sums in group order by the group in whose piece it starts.  Every group
// handles the same number of lookups whatever the run lengths are, and the summation order is a
// function of the data layout only (deterministic).
// Value of lookup (b,c):  drec[b, off_c + d]  +  g_fm[b] * (S[b,d] - rec[b, off_c + d])   [D-float row]
//                         g_lr[b]                                                          [D=1 row]
// the second term being d/de of 0.5 * sum_d((sum_f e)^2 - sum_f e^2) (inner_product.py:56-62).
// ---------------------------------------------------------------------------------------------
struct EmbFmBwdArgs {
    const float* drec;
    int64_t drec_ld;
    const float* rec;
    int64_t rec_ld;
    const float* S;
    const float* g_fm;
    const float* g_lr;
    const int64_t* col_out_off;
    const uint32_t* sorted_pos;
    const uint32_t* seg_start;
    const int32_t* n_unique;
    float* G;
    float* sq_partials;
    float* G1;
    float* sq1_partials;
    int32_t C, D, lanes_log2;
};

template <int VEC, bool FM, bool LR>
__device__ __forceinline__ void fx_lookup_value(const EmbFmBwdArgs& a, uint32_t p, int d0,
                                                bool lane_on, float (&v)[VEC], float& v1) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
    v1 = 0.f;
    if (p == 0xFFFFFFFFu) return;            // padding_idx / bad-id lookup: contributes nothing
    const uint32_t b = p / (uint32_t)a.C, c = p - b * (uint32_t)a.C;
    if constexpr (LR) v1 = a.g_lr[b];
    if (!lane_on) return;
    const int64_t off = a.col_out_off[c] + d0;
    if (a.drec) fx_load<VEC>(a.drec + (int64_t)b * a.drec_ld + off, v);
    if constexpr (FM) {
        float e[VEC], s[VEC];
        fx_load<VEC>(a.rec + (int64_t)b * a.rec_ld + off, e);
        fx_load<VEC>(a.S + (int64_t)b * a.D + d0, s);
        const float g = a.g_fm[b];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += g * (s[k] - e[k]);
    }
}

#define FX_BWD_INFL 4     // lookups in flight per lane group

template <int VEC, bool FM, bool LR>
__global__ __launch_bounds__(256) void k_emb_fm_bwd(EmbFmBwdArgs a) {
    // open pieces: F = a piece's first row started in an earlier piece, L = its last row goes on
    __shared__ float openF[256 * VEC], openL[256 * VEC];
    __shared__ float openF1[256], openL1[256];
    __shared__ int32_t rowF[256], rowL[256];          // indexed by lane group; -1 = none
    __shared__ int32_t wholeF[256];                   // 1: the piece is one row from end to end
    __shared__ uint32_t segs[257];
    __shared__ float red4[4];
    const int lanes = 1 << a.lanes_log2;
    const int NG = 256 >> a.lanes_log2;               // lane groups = unique rows per workgroup
    const int sub = threadIdx.x & (lanes - 1);
    const int g = threadIdx.x >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int nu = *a.n_unique;
    const int64_t u0 = (int64_t)blockIdx.x * NG;
    float sq = 0.f, sq1 = 0.f;
    if (u0 < nu) {                                     // block-uniform
        const int nrows = (int)((u0 + NG <= nu) ? NG : nu - u0);
        for (int t = threadIdx.x; t <= nrows; t += 256) segs[t] = a.seg_start[u0 + t];
        if (sub == 0) {
            rowF[g] = -1;
            rowL[g] = -1;
            wholeF[g] = 0;
        }
        __syncthreads();
        const uint32_t s0 = segs[0], s1 = segs[nrows];
        const uint32_t len = s1 - s0;
        const uint32_t lo = s0 + (uint32_t)(((uint64_t)len * (uint32_t)g) / (uint32_t)NG);
        const uint32_t hi = s0 + (uint32_t)(((uint64_t)len * (uint32_t)(g + 1)) / (uint32_t)NG);
        if (lo < hi) {
            // row of the first lookup of the piece: largest r with segs[r] <= lo
            int r = 0;
            {
                int l = 0, h = nrows;                  // segs[l] <= lo < segs[h]
                while (h - l > 1) {
                    const int mid = (l + h) >> 1;
                    if (segs[mid] <= lo) l = mid; else h = mid;
                }
                r = l;
            }
            bool first_open = lo > segs[r];
            float acc[VEC], acc1 = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            uint32_t i = lo;
            while (i < hi) {
                const uint32_t rend = segs[r + 1];
                const uint32_t end = rend < hi ? rend : hi;
                // lookups [i, end) of row r, FX_BWD_INFL at a time, summed in ascending order
                while (i < end) {
                    uint32_t p[FX_BWD_INFL];
                    float v[FX_BWD_INFL][VEC], v1[FX_BWD_INFL];
#pragma unroll
                    for (int j = 0; j < FX_BWD_INFL; ++j)
                        p[j] = (i + j < end) ? a.sorted_pos[i + j] : 0xFFFFFFFFu;
#pragma unroll
                    for (int j = 0; j < FX_BWD_INFL; ++j)
                        fx_lookup_value<VEC, FM, LR>(a, p[j], d0, lane_on, v[j], v1[j]);
#pragma unroll
                    for (int j = 0; j < FX_BWD_INFL; ++j) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) acc[k] += v[j][k];
                        acc1 += v1[j];
                    }
                    i = (i + FX_BWD_INFL < end) ? i + FX_BWD_INFL : end;
                }
                if (end == rend) {                     // row r ends inside this piece
                    if (first_open) {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) openF[k * 256 + threadIdx.x] = acc[k];
                        if (sub == 0) {
                            openF1[g] = acc1;
                            rowF[g] = r;
                        }
                        first_open = false;
                    } else {
                        if (lane_on) {
                            fx_store<VEC>(a.G + (u0 + r) * a.D + d0, acc);
#pragma unroll
                            for (int k = 0; k < VEC; ++k) sq = fmaf(acc[k], acc[k], sq);
                        }
                        if constexpr (LR) {
                            if (sub == 0) {
                                a.G1[u0 + r] = acc1;
                                sq1 = fmaf(acc1, acc1, sq1);
                            }
                        }
                    }
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
                    acc1 = 0.f;
                    ++r;
                } else {                               // the piece ends inside row r
                    if (first_open) {                  // ... and began inside it: one row, whole piece
#pragma unroll
                        for (int k = 0; k < VEC; ++k) openF[k * 256 + threadIdx.x] = acc[k];
                        if (sub == 0) {
                            openF1[g] = acc1;
                            rowF[g] = r;
                            wholeF[g] = 1;
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < VEC; ++k) openL[k * 256 + threadIdx.x] = acc[k];
                        if (sub == 0) {
                            openL1[g] = acc1;
                            rowL[g] = r;
                        }
                    }
                }
            }
        }
        __syncthreads();
        // a row spread over several pieces: the group whose piece holds its first lookups adds the
        // later pieces' partial sums in group order
        if (rowL[g] >= 0) {
            const int r = rowL[g];
            float tot[VEC], tot1 = openL1[g];
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] = openL[k * 256 + threadIdx.x];
            for (int h = g + 1; h < NG; ++h) {
                if (rowF[h] != r) {
                    if (rowF[h] < 0 && rowL[h] < 0) continue;      // an empty piece
                    break;
                }
                const int th = (h << a.lanes_log2) + sub;
#pragma unroll
                for (int k = 0; k < VEC; ++k) tot[k] += openF[k * 256 + th];
                tot1 += openF1[h];
                if (!wholeF[h]) break;
            }
            if (lane_on) {
                fx_store<VEC>(a.G + (u0 + r) * a.D + d0, tot);
#pragma unroll
                for (int k = 0; k < VEC; ++k) sq = fmaf(tot[k], tot[k], sq);
            }
            if constexpr (LR) {
                if (sub == 0) {
                    a.G1[u0 + r] = tot1;
                    sq1 = fmaf(tot1, tot1, sq1);
                }
            }
        }
    }
    // ||G||^2 of this workgroup's rows, fixed order (zero for workgroups past the last unique row)
    const float tot = fx_block_sum_256(sq, red4);
    if (threadIdx.x == 0) a.sq_partials[blockIdx.x] = tot;
    if constexpr (LR) {
        __syncthreads();
        const float tot1 = fx_block_sum_256(sq1, red4);
        if (threadIdx.x == 0) a.sq1_partials[blockIdx.x] = tot1;
    }
}

// launch 2: numeric weights (nn.Linear(1,D) per numeric feature, feature_embedding.py:153-154),
// their D=1 twins of LogisticRegression, and the LR bias.  Block j < Fd: feature j; block Fd: bias.
struct NumGradArgs {
    const float* drec;
    int64_t drec_ld;
    const float* rec;
    int64_t rec_ld;
    const float* S;
    const float* g_fm;
    const float* g_lr;
    const float* dense;
    int64_t dense_ld;
    const int64_t* num_out_off;
    float* dnum_w;
    float* dnum_w1;
    float* dbias1;
    int64_t B;
    int32_t Fd, D, Dp;
};

__global__ __launch_bounds__(1024) void k_emb_fm_numgrad(NumGradArgs a) {
    __shared__ float red[1024];
    __shared__ float red1[1024];
    const int j = blockIdx.x;
    const int d = threadIdx.x % a.Dp;
    const int grp = threadIdx.x / a.Dp;
    const int ngrp = 1024 / a.Dp;
    float acc = 0.f, acc1 = 0.f;
    if (j < a.Fd) {
        const int64_t off = a.num_out_off[j];
        for (int64_t b = grp; b < a.B; b += ngrp) {
            const float x = a.dense[b * a.dense_ld + j];
            if (d < a.D) {
                float v = a.drec ? a.drec[b * a.drec_ld + off + d] : 0.f;
                if (a.g_fm) v += a.g_fm[b] * (a.S[b * a.D + d] - a.rec[b * a.rec_ld + off + d]);
                acc = fmaf(x, v, acc);
            }
            if (d == 0 && a.g_lr) acc1 = fmaf(x, a.g_lr[b], acc1);
        }
    } else if (a.g_lr) {
        for (int64_t b = threadIdx.x; b < a.B; b += 1024) acc1 += a.g_lr[b];
    }
    red[threadIdx.x] = acc;
    red1[threadIdx.x] = acc1;
    __syncthreads();
    if (j < a.Fd) {
        for (int s = ngrp >> 1; s > 0; s >>= 1) {
            if (grp < s) {
                red[threadIdx.x] += red[threadIdx.x + s * a.Dp];
                red1[threadIdx.x] += red1[threadIdx.x + s * a.Dp];
            }
            __syncthreads();
        }
        if (grp == 0 && d < a.D) a.dnum_w[(int64_t)j * a.D + d] = red[d];
        if (threadIdx.x == 0 && a.dnum_w1) a.dnum_w1[j] = red1[0];
    } else {
        for (int s = 512; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red1[threadIdx.x] += red1[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0 && a.dbias1) a.dbias1[0] = red1[0];
    }
}

extern "C" int fx_emb_fm_bwd(const float* drec, int64_t drec_ld, const float* rec, int64_t rec_ld,
                             const float* S, const float* g_fm, const float* g_lr,
                             const int64_t* col_out_off, int32_t C, int32_t D,
                             const uint32_t* sorted_pos, const uint32_t* seg_start,
                             const int32_t* n_unique, int64_t n_max, float* G, float* sq_partials,
                             float* G1, float* sq1_partials, const float* dense, int64_t dense_ld,
                             const int64_t* num_out_off, int32_t Fd, int64_t B, float* dnum_w,
                             float* dnum_w1, float* dbias1, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256 && C >= 0 && Fd >= 0 && B >= 0, "fx_emb_fm_bwd: bad sizes");
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(g.lanes <= 64, "fx_emb_fm_bwd: D=%d needs %d lanes per row (max 64)", D, g.lanes);
    FX_CHECK_ARG(g_fm == nullptr || (rec && S), "fx_emb_fm_bwd: FM term without rec / S");
    FX_CHECK_ARG(drec != nullptr || g_fm != nullptr || g_lr != nullptr,
                 "fx_emb_fm_bwd: no upstream gradient at all");
    FX_CHECK_ARG((drec == nullptr || drec_ld % g.vec == 0) && (rec == nullptr || rec_ld % g.vec == 0),
                 "fx_emb_fm_bwd: leading dimensions not a multiple of %d", g.vec);
    hipStream_t s = fx_hip_stream(stream);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    if (C > 0 && n_max > 0) {
        FX_CHECK_ARG(col_out_off && sorted_pos && seg_start && n_unique && G && sq_partials,
                     "fx_emb_fm_bwd: null sparse argument");
        FX_CHECK_ARG(g_lr == nullptr || (G1 && sq1_partials), "fx_emb_fm_bwd: g_lr without G1");
        EmbFmBwdArgs a{drec, drec_ld, rec, rec_ld, S, g_fm, g_lr, col_out_off, sorted_pos, seg_start,
                       n_unique, G, sq_partials, G1, sq1_partials, C, D, ll};
        const int64_t blocks = fx_ceil_div(n_max, 256 / g.lanes);
        dim3 grid((unsigned)blocks);
#define FX_BWD_LAUNCH(V)                                                                          \
    do {                                                                                          \
        if (g_fm && g_lr) hipLaunchKernelGGL((k_emb_fm_bwd<V, true, true>), grid, dim3(256), 0, s, a);   \
        else if (g_fm) hipLaunchKernelGGL((k_emb_fm_bwd<V, true, false>), grid, dim3(256), 0, s, a);     \
        else if (g_lr) hipLaunchKernelGGL((k_emb_fm_bwd<V, false, true>), grid, dim3(256), 0, s, a);     \
        else hipLaunchKernelGGL((k_emb_fm_bwd<V, false, false>), grid, dim3(256), 0, s, a);              \
    } while (0)
        if (g.vec == 4) FX_BWD_LAUNCH(4);
        else if (g.vec == 2) FX_BWD_LAUNCH(2);
        else FX_BWD_LAUNCH(1);
#undef FX_BWD_LAUNCH
        FX_CHECK_LAUNCH();
    }
    if (Fd > 0 || (g_lr && dbias1)) {
        FX_CHECK_ARG(Fd == 0 || (dense && num_out_off && dnum_w), "fx_emb_fm_bwd: null numeric argument");
        int Dp = 1;
        while (Dp < D) Dp <<= 1;
        NumGradArgs na{drec, drec_ld, rec, rec_ld, S, g_fm, g_lr, dense, dense_ld, num_out_off,
                       dnum_w, g_lr ? dnum_w1 : nullptr, g_lr ? dbias1 : nullptr, B, Fd, D, Dp};
        const int nb = Fd + ((g_lr && dbias1) ? 1 : 0);
        hipLaunchKernelGGL(k_emb_fm_numgrad, dim3(nb), dim3(1024), 0, s, na);
        FX_CHECK_LAUNCH();
    }
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_sparse_adam_multi / fx_sparse_sgd_multi: the row update of every table group that shares one
// de-dup result (same uniq_row), one launch.
// ---------------------------------------------------------------------------------------------
struct MultiOptArgs {
    FxTableDev t[FX_MAX_TABLES];
    const uint32_t* uniq_row;
    const int32_t* n_unique;
    const fx_scalars* scal;
    int32_t n_tables, group_log2;
};

__device__ __forceinline__ float fx_reg_grad2(float p, float l1, float l2) {
    float r = l2 * p;
    if (l1 != 0.f) r += p > 0.f ? l1 : (p < 0.f ? -l1 : 0.f);
    return r;
}

template <int VEC, bool ADAM>
__device__ __forceinline__ void fx_update_row(const FxTableDev& t, int64_t u, int64_t row, int sub,
                                              const fx_scalars& sc) {
    const int lanes = 1 << t.lanes_log2;
    if (sub >= lanes) return;
    const int d0 = sub * VEC;
    if (d0 < t.D) {
        float p[VEC], g[VEC];
        const int64_t o = row * t.D + d0;
        fx_load<VEC>(t.table + o, p);
        fx_load<VEC>(t.G + u * t.D + d0, g);
        if (sc.reg_l1 != 0.f || sc.reg_l2 != 0.f) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] += fx_reg_grad2(p[k], sc.reg_l1, sc.reg_l2);
        }
        if constexpr (ADAM) {
            float m[VEC], v[VEC];
            fx_load<VEC>(t.m + o, m);
            fx_load<VEC>(t.v + o, v);
            const float w1 = 1.f - sc.beta1, w2 = 1.f - sc.beta2;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float gk = g[k] * sc.clip_coef;
                m[k] = m[k] + w1 * (gk - m[k]);                      // exp_avg.lerp_(grad, 1 - beta1)
                v[k] = fmaf(w2 * gk, gk, v[k] * sc.beta2);           // exp_avg_sq ... addcmul_
                const float denom = sqrtf(v[k]) / sc.bc2_sqrt + sc.eps;
                p[k] = p[k] - sc.step_size * (m[k] / denom);         // param.addcdiv_
            }
            fx_store<VEC>(t.m + o, m);
            fx_store<VEC>(t.v + o, v);
        } else {
            const float scale = sc.lr * sc.clip_coef;
#pragma unroll
            for (int k = 0; k < VEC; ++k) p[k] = p[k] - scale * g[k];
        }
        fx_store<VEC>(t.table + o, p);
    }
    if (sub == 0 && t.last_step) t.last_step[row] = sc.step;
}

template <bool ADAM>
__global__ __launch_bounds__(256) void k_sparse_update_multi(MultiOptArgs a) {
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t rpb = 256 >> a.group_log2;
    const int nu = *a.n_unique;
    const fx_scalars sc = *a.scal;
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.group_log2); u < nu;
         u += (int64_t)gridDim.x * rpb) {
        const int64_t row = a.uniq_row[u];
        for (int t = 0; t < a.n_tables; ++t) {
            const FxTableDev& tb = a.t[t];
            if (tb.vec == 4) fx_update_row<4, ADAM>(tb, u, row, sub, sc);
            else if (tb.vec == 2) fx_update_row<2, ADAM>(tb, u, row, sub, sc);
            else fx_update_row<1, ADAM>(tb, u, row, sub, sc);
        }
    }
}

static int fx_sparse_update_multi(bool adam, const fx_row_state* tables_host, int32_t n_tables,
                                  const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                                  const fx_scalars* scal, fx_stream_t stream, const char* who) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_MAX_TABLES, "%s: n_tables=%d not in [1,%d]", who,
                 n_tables, FX_MAX_TABLES);
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(tables_host && uniq_row && n_unique && scal, "%s: null pointer", who);
    MultiOptArgs a;
    memset(&a, 0, sizeof(a));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, a.t, &gl, who, false);
    if (st != FX_OK) return st;
    for (int t = 0; t < n_tables; ++t) {
        FX_CHECK_ARG(a.t[t].G != nullptr, "%s: table %d has no gradient", who, t);
        FX_CHECK_ARG(!adam || (a.t[t].m && a.t[t].v), "%s: table %d has no Adam moments", who, t);
    }
    a.uniq_row = uniq_row;
    a.n_unique = n_unique;
    a.scal = scal;
    a.n_tables = n_tables;
    a.group_log2 = gl;
    int64_t blocks = fx_ceil_div(n_max, 256 >> gl);
    if (blocks > 256 * 64) blocks = 256 * 64;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (adam) hipLaunchKernelGGL(k_sparse_update_multi<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_sparse_update_multi<false>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_sparse_adam_multi(const fx_row_state* tables_host, int32_t n_tables,
                                    const uint32_t* uniq_row, const int32_t* n_unique,
                                    int64_t n_max, const fx_scalars* scal, fx_stream_t stream) {
    return fx_sparse_update_multi(true, tables_host, n_tables, uniq_row, n_unique, n_max, scal,
                                  stream, "fx_sparse_adam_multi");
}

extern "C" int fx_sparse_sgd_multi(const fx_row_state* tables_host, int32_t n_tables,
                                   const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                                   const fx_scalars* scal, fx_stream_t stream) {
    return fx_sparse_update_multi(false, tables_host, n_tables, uniq_row, n_unique, n_max, scal,
                                  stream, "fx_sparse_sgd_multi");
}

// ---------------------------------------------------------------------------------------------
// fx_pack_columns_multi: like fx_pack_columns, but every column names its own destination matrix,
// so the id block, the numeric block and the label of one batch are cast in ONE launch.
// ---------------------------------------------------------------------------------------------
#define FX_PACKM_MAX_COLS 96
struct PackMultiArgs {
    const void* col[FX_PACKM_MAX_COLS];
    void* out[FX_PACKM_MAX_COLS];          // first element of the column's destination
    int32_t ld[FX_PACKM_MAX_COLS];
    int16_t width[FX_PACKM_MAX_COLS];
    int8_t dtype[FX_PACKM_MAX_COLS];
    int8_t out_dtype[FX_PACKM_MAX_COLS];
    int64_t B;
};

__global__ __launch_bounds__(256) void k_pack_columns_multi(PackMultiArgs a) {
    const int c = blockIdx.y;
    const int64_t w = a.width[c];
    const int64_t n = a.B * w;
    const int dt = a.dtype[c];
    const void* src = a.col[c];
    const int64_t ld = a.ld[c];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / w, k = i - b * w;
        if (a.out_dtype[c] == FX_I32) {
            int32_t v;
            if (dt == FX_F64) v = (int32_t) reinterpret_cast<const double*>(src)[i];
            else if (dt == FX_I64) v = (int32_t) reinterpret_cast<const int64_t*>(src)[i];
            else if (dt == FX_F32) v = (int32_t) reinterpret_cast<const float*>(src)[i];
            else v = reinterpret_cast<const int32_t*>(src)[i];
            reinterpret_cast<int32_t*>(a.out[c])[b * ld + k] = v;
        } else {
            float v;
            if (dt == FX_F64) v = (float) reinterpret_cast<const double*>(src)[i];
            else if (dt == FX_I64) v = (float) reinterpret_cast<const int64_t*>(src)[i];
            else if (dt == FX_F32) v = reinterpret_cast<const float*>(src)[i];
            else v = (float) reinterpret_cast<const int32_t*>(src)[i];
            reinterpret_cast<float*>(a.out[c])[b * ld + k] = v;
        }
    }
}

extern "C" int fx_pack_columns_multi(const void* const* cols_host, const int32_t* dtypes_host,
                                     const int32_t* widths_host, void* const* outs_host,
                                     const int32_t* out_dtypes_host, const int64_t* out_lds_host,
                                     int32_t ncols, int64_t B, fx_stream_t stream) {
    FX_CHECK_ARG(ncols >= 0 && ncols <= FX_PACKM_MAX_COLS,
                 "fx_pack_columns_multi: ncols=%d not in [0,%d]", ncols, FX_PACKM_MAX_COLS);
    FX_CHECK_ARG(B >= 0, "fx_pack_columns_multi: B < 0");
    if (ncols == 0 || B == 0) return FX_OK;
    FX_CHECK_ARG(cols_host && dtypes_host && widths_host && outs_host && out_dtypes_host &&
                     out_lds_host, "fx_pack_columns_multi: null pointer");
    PackMultiArgs a;
    memset(&a, 0, sizeof(a));
    int64_t maxw = 1;
    for (int c = 0; c < ncols; ++c) {
        FX_CHECK_ARG(cols_host[c] && outs_host[c], "fx_pack_columns_multi: column %d is null", c);
        FX_CHECK_ARG(dtypes_host[c] >= FX_F32 && dtypes_host[c] <= FX_I64,
                     "fx_pack_columns_multi: bad dtype %d for column %d", dtypes_host[c], c);
        FX_CHECK_ARG(out_dtypes_host[c] == FX_I32 || out_dtypes_host[c] == FX_F32,
                     "fx_pack_columns_multi: out dtype of column %d must be FX_I32 or FX_F32", c);
        FX_CHECK_ARG(widths_host[c] >= 1 && widths_host[c] <= 32767 &&
                         out_lds_host[c] >= widths_host[c] && out_lds_host[c] < (1ll << 31),
                     "fx_pack_columns_multi: bad width / ld of column %d", c);
        a.col[c] = cols_host[c];
        a.out[c] = outs_host[c];
        a.ld[c] = (int32_t)out_lds_host[c];
        a.width[c] = (int16_t)widths_host[c];
        a.dtype[c] = (int8_t)dtypes_host[c];
        a.out_dtype[c] = (int8_t)out_dtypes_host[c];
        if (widths_host[c] > maxw) maxw = widths_host[c];
    }
    a.B = B;
    int64_t gx = fx_ceil_div(B * maxw, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_pack_columns_multi, dim3((unsigned)gx, (unsigned)ncols), dim3(256), 0,
                       fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
