/*
 * fxctr.h — C-ABI of the MI355X-native CTR hot path (libfxctr.so, gfx950 only).
 *
 * Drop-in boundary.  The reference (reczoo/FuxiCTR) has no FFI: its hot path is a Python class
 * API (fuxictr.pytorch FeatureEmbedding / BaseModel) that dispatches to stock ATen ops.  Every
 * entry point below replaces the ATen work issued by the reference lines cited next to it
 * (paths relative to the reference checkout).  The Python shim in fuxictr_amd/ binds these with
 * ctypes and mirrors the reference class API on top.
 *
 * Conventions
 *   - plain C types only: device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - every pointer is a DEVICE pointer unless the name ends in _host.
 *   - every function returns FX_OK (0) or an FX_ERR_* code; fx_last_error() gives the message
 *     (thread-local).  No exceptions cross the ABI.  The caller owns all memory.
 *   - all launches are asynchronous on `stream`; nothing here synchronises, allocates or frees,
 *     so every call is hipGraph-capturable.  Per-step dynamic scalars (step, lr, clip
 *     coefficient, unique-row count) live in device memory for the same reason.
 *   - matrices are row-major fp32; "ld" = leading dimension (row stride) in elements.
 */
#ifndef FXCTR_H
#define FXCTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FX_ABI_VERSION 1

typedef void* fx_stream_t; /* hipStream_t; NULL = the null stream */

enum { FX_OK = 0, FX_ERR_INVALID = 1, FX_ERR_HIP = 2, FX_ERR_UNSUPPORTED = 3 };
enum { FX_F32 = 0, FX_F64 = 1, FX_I32 = 2, FX_I64 = 3, FX_BF16 = 4 };

/* bits of fx_scalars.err_flag (sticky, set by kernels, read by the host at its own sync points) */
enum { FX_FLAG_BAD_ID = 1, FX_FLAG_A2A_OVERFLOW = 2 };

/* Device-resident per-step scalars (64 bytes; the shim allocates it as 16 x 4-byte words).
 * Mirrors the python-side scalars of torch.optim.Adam._single_tensor_adam and
 * torch.nn.utils.clip_grad_norm_ as used by rank_model.py:321-322. */
typedef struct fx_scalars {
    int32_t step;      /*  0: optimizer step t, 1-based after fx_opt_begin_step */
    int32_t err_flag;  /*  1 */
    float lr;          /*  2: written by the host when lr changes (rank_model.py:221-234) */
    float beta1;       /*  3 */
    float beta2;       /*  4 */
    float eps;         /*  5 */
    float bc1;         /*  6: 1 - beta1^t */
    float bc2_sqrt;    /*  7: sqrt(1 - beta2^t) */
    float step_size;   /*  8: lr / bc1 */
    float clip_coef;   /*  9: min(1, max_norm / (total_norm + 1e-6)) */
    float total_norm;  /* 10 */
    float max_norm;    /* 11: <= 0 disables clipping (coef = 1) */
    float loss;        /* 12: scratch for the fused loss kernel */
    float reg_l1;      /* 13: embedding regularizer, l1 weight (0 = off)   rank_model.py:106-112 */
    float reg_l2;      /* 14: embedding regularizer, l2 weight: loss += l1 |p|_1 + l2/2 |p|_2^2 */
    int32_t series_tcap; /* 15: 0, or the entry count of the Adam series table that FOLLOWS this struct in the
                          *     same allocation (fx_adam_series_build); the exact-mode catch-up kernels then
                          *     sum a row's missed zero-gradient steps from it instead of replaying them */
} fx_scalars;

int fx_abi_version(void);
const char* fx_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Input packing.  Replaces the per-feature `.long()` / `.float().view(-1,1)` casts of
 * fuxictr/pytorch/layers/embeddings/feature_embedding.py:280-291 (39 tiny ATen casts per
 * forward at Criteo shape) by one launch: ncols device columns, column c being a contiguous
 * [B, widths[c]] array of dtype dtypes[c], are converted and written side by side into
 * out[b*out_ld + out_col0 + (running column offset)].  out_dtype is FX_I32 (ids) or FX_F32.
 * cols_host / dtypes_host / widths_host are HOST arrays (<= 64 columns per call).
 * ------------------------------------------------------------------------------------------ */
int fx_pack_columns(const void* const* cols_host, const int32_t* dtypes_host,
                    const int32_t* widths_host, int32_t ncols, int64_t B, int32_t out_dtype,
                    void* out, int64_t out_ld, int64_t out_col0, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-field embedding lookup, forward.  Replaces FeatureEmbeddingDict.forward +
 * dict2tensor (feature_embedding.py:261-297, :230-259): C id-columns gather D-float rows from
 * one packed table, Fd numeric columns are expanded by their nn.Linear(1,D,bias=False) weight
 * (feature_embedding.py:153-154, :280-282), everything is written straight into its final slot
 * of the per-sample output record (no stack/cat pass):
 *     out[b*out_ld + col_out_off[c] + d] = table[(col_row_base[c] + ids[b,c]) * table_ld + d]
 *     out[b*out_ld + num_out_off[j] + d] = dense[b,j] * num_w[j*D + d]
 * ids outside [0, col_vocab[c]) set FX_FLAG_BAD_ID in scal->err_flag and read as a zero row.
 * table_ld: row stride of `table` in floats (<= 0: D, a packed table; W for a table that is the
 * first field of a [p | m | v | last_step] row record, see fx_row_state).
 * ------------------------------------------------------------------------------------------ */
int fx_emb_gather_fwd(const float* table, int32_t D, const int32_t* ids, int64_t ids_ld,
                      const int64_t* col_row_base, const int32_t* col_vocab,
                      const int64_t* col_out_off, int32_t C, const float* dense,
                      int64_t dense_ld, const float* num_w, const int64_t* num_out_off,
                      int32_t Fd, float* out, int64_t out_ld, int64_t B, fx_scalars* scal,
                      int64_t table_ld, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Pooled sequence features, forward.  Replaces the lookup of a sequence feature followed by its
 * `feature_encoder` (feature_embedding.py:283-295) when that encoder is MaskedSumPooling or
 * MaskedAveragePooling (pooling.py:59-70, :32-47) — the default encoder of every sequence feature
 * (feature_processor.py:379).  Sequence s occupies id columns seq_col0[s] .. +seq_len[s] of `ids`
 * (indices into col_row_base / col_vocab); its [seq_len, D] history is reduced in registers and
 * only the pooled row is written:
 *     sum[b,s,:]  = sum_l table[(col_row_base[c] + ids[b,c]) * D + :],  c = seq_col0[s] + l
 *     denom[b,s]  = #{l : sum_d row_l[d] != 0} + 1e-12          (the reference's inferred mask)
 *     out[b*out_ld + seq_out_off[s] + :] = sum            (seq_mode[s] == FX_POOL_SUM)
 *                                        = sum / denom    (seq_mode[s] == FX_POOL_MEAN)
 * denom [B, n_seq] is kept for the backward (fx_emb_grad_reduce_scaled).  Bad ids as in
 * fx_emb_gather_fwd.  Rows of more than 64 lanes (D > 256 / 128 / 64 for D % 4 / % 2 / odd) are
 * rejected.
 * ------------------------------------------------------------------------------------------ */
enum { FX_POOL_SUM = 0, FX_POOL_MEAN = 1 };
int fx_emb_seq_pool_fwd(const float* table, int32_t D, const int32_t* ids, int64_t ids_ld,
                        const int64_t* col_row_base, const int32_t* col_vocab,
                        const int32_t* seq_col0, const int32_t* seq_len, const int32_t* seq_mode,
                        const int64_t* seq_out_off, int32_t n_seq, float* out, int64_t out_ld,
                        float* denom, int64_t B, fx_scalars* scal,
                        int64_t table_ld /* row stride in floats; <= 0: D */, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Index de-duplication for the sparse backward/update.  Builds, for the B*C lookups of a batch,
 * the list of unique packed-table rows and the (stable) sorted lookup positions of each:
 *     key(b,c) = col_row_base[c] + ids[b,c]       (sentinel total_rows for padding_idx / bad ids)
 *     sorted_key/sorted_pos = stable radix sort of (key, pos=b*C+c) by key
 *     uniq_row[u], seg_start[u]..seg_start[u+1] = the u-th distinct key and its run in sorted_*
 *     *n_unique = number of distinct non-sentinel keys
 *     sorted_uid[i] (optional, may be NULL) = u of sorted lookup i, 0xFFFFFFFF for sentinels
 * n_shards = 1 for an unsharded table (see fx_shard_plan for n_shards > 1).
 * columns_sorted != 0 (hint; needs n_shards = 1): every id column owns its own table and
 * col_row_base is strictly increasing — each column is then sorted by one workgroup entirely in
 * LDS (a hand-written stable radix sort, 8 bits a pass) over only the bits its vocabulary needs
 * (B <= 8192; larger batches use the generic path: key build + device-wide LSD radix sort + a two-launch
 * scan / scatter of the unique rows — all own kernels, no library).  In that mode padding_idx / bad-id lookups stay inside their column's key range with
 * sorted_pos = 0xFFFFFFFF ("contributes nothing"): the padding row may appear as a unique row
 * whose reduced gradient is exactly zero.
 * columns_sorted == 2 (hint; needs n_shards = 1, B*C <= 2^21): the caller does not need ASCENDING unique rows,
 * only every row's lookups side by side in position order and a deterministic order of the rows — sequence
 * columns that alias their target's table (DIN), shared tables, B > 8192.  The rows are hashed into 256
 * buckets by their low 8 bits (one stable partition pass) and every bucket is sorted by one workgroup in LDS
 * (csrc/fx_dedup_lds.hip; a bucket of more than 8192 lookups is sorted through global memory by the same
 * code): 4 launches instead of the generic path's 10.  sorted_key / uniq_row come out ordered by
 * (row & 255, row >> 8); padding / bad-id lookups form the tail (key = total_rows, sorted_pos = sorted_uid =
 * 0xFFFFFFFF).  Everything else as in the generic path.
 * This replaces the zero-filled dense [V,D] gradient + index_add of aten::embedding_dense_backward
 * (triggered at rank_model.py:320) — no dense gradient ever exists.  Deterministic.
 * Packed tables are limited to < 2^32 - 1 rows.
 * ------------------------------------------------------------------------------------------ */
size_t fx_dedup_workspace_bytes(int64_t n_lookups);
int fx_dedup(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
             const int64_t* col_row_base, const int32_t* col_vocab, const int32_t* col_pad,
             int64_t total_rows, void* workspace, size_t workspace_bytes, uint32_t* sorted_key,
             uint32_t* sorted_pos, uint32_t* uniq_row, uint32_t* seg_start, int32_t* n_unique,
             uint32_t* sorted_uid, int32_t n_shards, int32_t columns_sorted,
             fx_scalars* begin_scal /* or NULL: fx_opt_begin_step fused into the first launch */,
             fx_stream_t stream);

/* The same de-dup for keys that arrive as n_runs consecutive runs of run_len ids of ONE table
 * (rows [0, vocab), `pad` = the id that means "nothing"), each run ascending with its pad entries at
 * the tail — what the owner of a row-sharded table receives from its peers (fx_shard_plan's
 * send_idx).  A stable R-way merge by rank counting replaces the sort; outputs exactly as fx_dedup
 * (sorted_pos = index into ids; sentinel key = vocab).  Runs that are not ascending give an
 * unspecified (but memory-safe) result.  Workspace: fx_dedup_workspace_bytes(n_runs * run_len). */
int fx_dedup_sorted_runs(const int32_t* ids, int32_t n_runs, int64_t run_len, int32_t vocab,
                         int32_t pad, void* workspace, size_t workspace_bytes,
                         uint32_t* sorted_key, uint32_t* sorted_pos, uint32_t* uniq_row,
                         uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                         fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Row-sharded tables (new functionality: the reference has no multi-GPU path).
 * With the packed table sharded row-wise over n_shards ranks (owner = row % n_shards, local row =
 * row / n_shards, rows_per_shard = ceil(total_rows / n_shards)) fx_dedup(n_shards > 1) keys the
 * lookups owner-major, so the unique keys of a batch are already grouped by owning rank.
 * fx_shard_plan turns them into fixed-capacity all-to-all buffers (no host-side counts):
 *     send_idx[o*cap + j]   local row (at owner o) of this rank's j-th unique key owned by o,
 *                           rows_per_shard (the owner's all-zero pad row) past the bucket end
 *     uniq_slot[u]          slot o*cap + j of unique key u in the padded buffers
 *     lookup_slot[b*C + c]  slot of lookup (b,c): the id matrix to gather from the received rows
 *                           (n_shards*cap = pad slot for padding_idx lookups)
 * A bucket larger than cap sets FX_FLAG_A2A_OVERFLOW.  sorted_uid is fx_dedup's optional output
 * (unique index of every sorted lookup).  fx_scatter_rows moves reduced gradient rows (row stride
 * dst_ld: several table groups may share one exchange block) into their
 * all-to-all slots; fx_sum_parts reduces partial squared norms to one device scalar (the
 * rank-local table term of the global clip norm, summed across ranks by the host's all-reduce).
 * ------------------------------------------------------------------------------------------ */
/* global_keys = 0: uniq_key are owner-major keys (fx_dedup with n_shards = N).
 * global_keys = 1: uniq_key are GLOBAL packed rows in ascending order (fx_dedup with n_shards = 1,
 *   e.g. its column fast path): owner = g % N, local row = g / N, an owner's bucket keeps ascending
 *   row order; needs `workspace` of fx_shard_plan_workspace_ints(n_lookups, n_shards) int32 words;
 *   positions whose id was padding / out of range get the pad slot. */
int64_t fx_shard_plan_workspace_ints(int64_t n_lookups, int32_t n_shards);
/*   slot_uniq (global_keys = 1 only; may be NULL): the inverse map, slot_uniq[o*cap + j] = u of the
 *   unique key in that slot, -1 for empty slots — fx_fill_grad_block writes the gradient exchange block
 *   by it.  Three launches (histograms; slots + bucket tails + fills; lookup slots). */
int fx_shard_plan(const uint32_t* uniq_key, const int32_t* n_unique, const uint32_t* sorted_pos,
                  const uint32_t* sorted_uid, int64_t n_lookups, int32_t n_shards,
                  int64_t total_rows, int32_t cap, int32_t* send_idx, int32_t* uniq_slot,
                  int32_t* lookup_slot, fx_scalars* scal, int32_t global_keys, int32_t* workspace,
                  int32_t* slot_uniq, fx_stream_t stream);
/* Gradient exchange of the row-sharded backward, one launch each side.
 * fx_fill_grad_block (requester): block[e, off_t .. off_t + D_t) = G_t[slot_uniq[e], :] for every slot
 *   e < n_slots of the [n_slots, ld] exchange block, zeros for empty slots (slot_uniq < 0), for table
 *   groups without a gradient this step (G_host[t] = NULL) and for pad columns: the block is written
 *   once, densely.  G_host / D_host / off_host are HOST arrays (<= 4 table groups).
 * fx_owner_grad_reduce (owner): for every unique owned row u of the owner-side de-dup (sorted_pos /
 *   seg_start / n_unique of fx_dedup_sorted_runs over the received ids), G_t[u, :] = sum over the row's
 *   run — one entry per requesting rank, ascending rank order — of grecv[sorted_pos[i], off_t + :], for
 *   every table group in one launch, and sq_partials[b] = sum of G^2 of all groups over rows
 *   [32 b, 32 b + 32) in a fixed order (fx_owner_grad_reduce_partials(n_max) entries; the table part of
 *   clip_grad_norm_, rank_model.py:321).  Replaces aten::embedding_dense_backward's accumulation
 *   across what would be one device in the reference (rank_model.py:320). */
int fx_fill_grad_block(const float* const* G_host, const int32_t* D_host, const int32_t* off_host,
                       int32_t n_tables, const int32_t* slot_uniq, int64_t n_slots, float* block,
                       int64_t ld, fx_stream_t stream);
int64_t fx_owner_grad_reduce_partials(int64_t n_max);
int fx_owner_grad_reduce(const float* grecv, int64_t ld, const uint32_t* sorted_pos,
                         const uint32_t* seg_start, const int32_t* n_unique, int64_t n_max,
                         float* const* G_host, const int32_t* D_host, const int32_t* off_host,
                         int32_t n_tables, float* sq_partials, fx_stream_t stream);
int fx_scatter_rows(const float* src, const int32_t* row_map, const int32_t* n_rows, int64_t n_max,
                    int32_t D, float* dst, int64_t dst_ld, fx_stream_t stream);
/* The received row block of an exchange, [n_rows, src_ld] with one column range (off, width) per
 * table group, into one contiguous [n_rows + zero_tail_rows, width] buffer per group; the tail rows
 * (the pad slot) are zeroed.  dst_host / off_host / width_host are HOST arrays (<= 4 parts). */
int fx_split_rows(const float* src, int64_t src_ld, int64_t n_rows, int32_t n_parts,
                  float* const* dst_host, const int32_t* off_host, const int32_t* width_host,
                  int32_t zero_tail_rows, fx_stream_t stream);
int fx_sum_parts(const float* const* parts_host, const int64_t* counts_host, int32_t n_parts,
                 float* out, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Embedding backward, sparse side: G[u,:] = sum over the run of unique row u of
 * dout[b*dout_ld + col_out_off[c] + :], summed in a fixed order (deterministic; runs longer than
 * 32 lookups — hot rows of tiny tables — are reduced by a whole workgroup each).
 * Also writes per-block partial sums of ||G||^2 into sq_partials[0..n_partials), n_partials =
 * fx_emb_grad_reduce_partials(n_max, D); rows >= *n_unique are not touched.
 * scratch: fx_emb_grad_reduce_scratch_ints(n_max) int32 words of device scratch; word 0 must be 0
 * on entry (allocate zeroed once) and is 0 again on return, so the buffer can be reused as is.
 * ------------------------------------------------------------------------------------------ */
int64_t fx_emb_grad_reduce_partials(int64_t n_max, int32_t D);
int64_t fx_emb_grad_reduce_scratch_ints(int64_t n_max);
int fx_emb_grad_reduce(const float* dout, int64_t dout_ld, const int64_t* col_out_off, int32_t C,
                       int32_t D, const uint32_t* sorted_pos, const uint32_t* seg_start,
                       const int32_t* n_unique, int64_t n_max, float* G, float* sq_partials,
                       int32_t* scratch, fx_stream_t stream);
/* The same with a per-lookup divisor: the lookup (b, c) contributes
 * dout[b*dout_ld + col_out_off[c] + :] / denom[b*denom_ld + col_denom[c]] when col_denom[c] >= 0
 * (autograd of the mean pooling of fx_emb_seq_pool_fwd: every position of the history receives
 * dpooled / denom), and the plain value when col_denom[c] < 0.  All id columns of a pooled
 * sequence carry the pooled slot's offset in col_out_off. */
int fx_emb_grad_reduce_scaled(const float* dout, int64_t dout_ld, const int64_t* col_out_off,
                              const int32_t* col_denom, const float* denom, int64_t denom_ld,
                              int32_t C, int32_t D, const uint32_t* sorted_pos,
                              const uint32_t* seg_start, const int32_t* n_unique, int64_t n_max,
                              float* G, float* sq_partials, int32_t* scratch, fx_stream_t stream);

/* Numeric-feature weight gradient: dnum_w[j,d] = sum_b dense[b,j] * dout[b*dout_ld + num_out_off[j] + d]
 * (autograd of the nn.Linear(1,D) at feature_embedding.py:280-282).  Deterministic tree sum. */
#define FX_NUMGRAD_CHUNKS 16
int fx_emb_numeric_grad(const float* dout, int64_t dout_ld, const int64_t* num_out_off,
                        const float* dense, int64_t dense_ld, int32_t Fd, int32_t D, int64_t B,
                        float* dnum_w, float* workspace /* FX_NUMGRAD_CHUNKS*Fd*D floats or NULL */,
                        fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Optimizer scalars.  fx_opt_begin_step: step += 1 and the bias corrections of
 * torch.optim.Adam (bias_correction1/2, step_size; computed in double like the Python side).
 * fx_clip_coef: total_norm = sqrt(sum of all given partial arrays), clip_coef as
 * torch.nn.utils.clip_grad_norm_ (rank_model.py:321).  parts_host: HOST array of n_parts device
 * pointers, counts_host their lengths (<= 16 arrays).
 * ------------------------------------------------------------------------------------------ */
int fx_opt_begin_step(fx_scalars* scal, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Adam series table (exact mode; torch.optim.Adam as stepped at rank_model.py:322 over rows whose
 * gradient is zero, torch_utils.py:72-76).  Step i after a row's last update t moves an element by
 *     u_i = lr m b1^i / (1 - b1^(t+i)) / (sqrt(v) b2^(i/2) / sqrt(1 - b2^(t+i)) + eps)
 *         = (lr m / sqrt(v)) * w_i(t) / (g_i(t) + c),      c = eps / sqrt(v),
 * w_i, g_i the same for every element of every row last updated at t.  The table holds, per t, the
 * infinite sum F(t; c) = sum_{i>=1} w_i / (g_i + c) as a few segments of i, each expanded about its
 * weighted mean g:  sum_seg = y (c0 + y^2 (c2 + y (c3 + y (c4 + y (c5 + y c6))))), y = g / (g + c)
 * (uniformly convergent in c >= 0).  The k missed steps of a row are then
 *     sum_{i<=k} u_i = (lr m / sqrt(v)) * ( F(t; c) - (b1/sqrt(b2))^k F(t + k; c b2^(-k/2)) ),
 * k <= FX_SERIES_KDIR steps are still replayed one by one (the difference would cancel).
 * Layout after the 16 words of fx_scalars: 16 header words | 128 entries x 8 segments x 8 floats
 * (t < 128; word 7 of an entry = its segment count) | (tcap - 128) entries x 8 floats (one segment);
 * t >= tcap reads entry tcap - 1 (both bias corrections are 1 in fp32 there).
 * fx_adam_series_words(tcap): 4-byte words the table needs behind the struct.
 * fx_adam_series_build: reads beta1 / beta2 from scal, fills the table (fp64 inside, fp32 out), writes
 * the largest relative error of any entry against the directly summed series (probed over c from 0 to
 * 100 g_1) to header word 2 and sets scal->series_tcap = tcap.  The host reads that word once and clears
 * series_tcap if it is above its tolerance (other betas than torch's defaults may not converge).
 * ------------------------------------------------------------------------------------------ */
#define FX_SERIES_KDIR 12
#define FX_SERIES_EARLY 128
int64_t fx_adam_series_words(int32_t tcap);
int fx_adam_series_build(fx_scalars* scal, int32_t tcap, fx_stream_t stream);
int fx_clip_coef(const float* const* parts_host, const int64_t* counts_host, int32_t n_parts,
                 fx_scalars* scal, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Sparse-row optimizers on the packed table (replace the dense torch.optim step over every row,
 * rank_model.py:322 / torch_utils.py:76).  All take the unique rows of fx_dedup and the reduced
 * gradient rows G (scaled by scal->clip_coef inside).
 *  fx_sparse_adam   : p,m,v of the touched rows, Adam step t = scal->step; last_step[row] = t.
 *  fx_adam_catchup  : "exact" mode — dense Adam also moves rows whose gradient is zero (their
 *                     m/(sqrt(v)+eps) is non-zero once touched).  Replays the zero-gradient
 *                     steps last_step[row]+1 .. upto for the given rows (uniq_row != NULL) or for
 *                     all rows [0,total_rows) (uniq_row == NULL, flush before evaluate/save), so
 *                     the table equals the reference's dense-Adam table.  upto_offset is added
 *                     to scal->step (-1: bring rows to t-1 before the forward of step t; 0: flush).
 *  fx_sparse_sgd    : p -= lr * clip * g (exactly the dense SGD result); last_step (nullable) is
 *                     marked like fx_sparse_adam's.
 * With scal->reg_l1/reg_l2 set, both add the regularizer gradient r(p) = l1 sign(p) + l2 p to g.
 * ------------------------------------------------------------------------------------------ */
int fx_sparse_adam(float* table, float* m, float* v, int32_t* last_step, int32_t D,
                   const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                   const float* G, const fx_scalars* scal, fx_stream_t stream);
int fx_adam_catchup(float* table, float* m, float* v, int32_t* last_step, int32_t D,
                    const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                    int64_t total_rows, int32_t upto_offset, const fx_scalars* scal,
                    fx_stream_t stream);
int fx_sparse_sgd(float* table, int32_t* last_step, int32_t D, const uint32_t* uniq_row,
                  const int32_t* n_unique, int64_t n_max, const float* G, const fx_scalars* scal,
                  fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * embedding_regularizer != 0 (BaseModel.regularization_loss, rank_model.py:95-112): the loss gets
 * sum over every FeatureEmbeddingDict parameter of (l1 |p|_1 + l2/2 |p|_2^2), so EVERY table row
 * has a gradient each step — the reference's dense pass, kept dense here (but fused: 2 reads +
 * 1 read-modify-write of the table per step instead of autograd's zero-fill + norm + clip + Adam).
 *  fx_reg_stats        : partials[0..NB) = sum p^2, [NB..2NB) = sum |p|, [2NB..3NB) = sum r(p)^2
 *                        over x[0..n)  (NB = FX_REG_BLOCKS; fixed order -> deterministic)
 *  fx_reg_cross        : partials[0..FX_REG_CROSS_BLOCKS) = sum 2 G.r(p) over the touched rows;
 *                        |G + r|^2 over all rows = sum r^2 + sum G^2 + this  (for the global clip)
 *  fx_reg_dense_update : Adam (adam=1) / SGD (adam=0) step with g = r(p) * clip for every row
 *                        whose last_step != scal->step (touched rows were stepped, with G + r, by
 *                        fx_sparse_adam / fx_sparse_sgd, which mark last_step).
 * ------------------------------------------------------------------------------------------ */
#define FX_REG_BLOCKS 1024
#define FX_REG_CROSS_BLOCKS 256
int fx_reg_stats(const float* x, int64_t n, const fx_scalars* scal, float* partials,
                 fx_stream_t stream);
int fx_reg_cross(const float* table, int32_t D, const uint32_t* uniq_row, const int32_t* n_unique,
                 int64_t n_max, const float* G, const fx_scalars* scal, float* partials,
                 fx_stream_t stream);
int fx_reg_dense_update(float* table, float* m, float* v, const int32_t* last_step,
                        int64_t total_rows, int32_t D, int32_t adam, const fx_scalars* scal,
                        fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense (multi-tensor) side of clip + Adam/SGD for the MLP / CrossNet / bias parameters.
 * ptr arrays are HOST arrays of n device pointers (<= 64 per call), sizes_host element counts.
 *  fx_mt_sqnorm : sq_partials[i*FX_MT_BLOCKS + k] = partial ||g_i||^2   (FX_MT_BLOCKS per tensor)
 *  fx_mt_adam   : torch.optim.Adam single-tensor formulas with g * clip_coef
 *  fx_mt_sgd    : p -= lr * clip_coef * g
 * ------------------------------------------------------------------------------------------ */
#define FX_MT_BLOCKS 96
int fx_mt_sqnorm(const float* const* grads_host, const int64_t* sizes_host, int32_t n,
                 float* sq_partials, fx_stream_t stream);
int fx_mt_adam(float* const* params_host, const float* const* grads_host, float* const* m_host,
               float* const* v_host, const int64_t* sizes_host, int32_t n, const fx_scalars* scal,
               fx_stream_t stream);
int fx_mt_sgd(float* const* params_host, const float* const* grads_host,
              const int64_t* sizes_host, int32_t n, const fx_scalars* scal, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * FM second-order term + LR first-order term (DeepFM / FM / xDeepFM).
 *  fx_fm_fwd : out[b] = 0.5 * sum_d ((sum_f e[b,f,d])^2 - sum_f e[b,f,d]^2) (+ addend[b])
 *              (fuxictr/pytorch/layers/interactions/inner_product.py:55-62, "product_sum";
 *              the addend fuses factorization_machine.py:58).
 *  fx_fm_bwd : demb[b,f,d] (+)= g[b] * (sum_f e[b,f,d] - e[b,f,d])   (accumulate != 0: +=)
 *  fx_lr_fwd : out[b] = sum_c table1[col_row_base[c] + ids[b,c]] + sum_j dense[b,j]*num_w1[j] + bias
 *              (logistic_regression.py:55-58: a D=1 FeatureEmbedding summed over fields).
 * ------------------------------------------------------------------------------------------ */
int fx_fm_fwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D, const float* addend,
              float* out, int64_t B, fx_stream_t stream);
int fx_fm_bwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D, const float* g,
              float* demb, int64_t demb_ld, int32_t accumulate, int64_t B, fx_stream_t stream);
/* DLRM "dot" interaction: out[b, p] = <e[b,i,:], e[b,j,:]> for the pairs i < j in row-major
 * upper-triangle order (inner_product.py:63-66: bmm + triu masked_select), and its backward.
 * Rows of out / g have out_ld / g_ld floats.  tail (0, or >= D): the LAST field is also appended to the
 * products (DLRM.py:117-120 concatenates the bottom tower's vector): out[b, P .. P+D) = e[b, F-1, :],
 * out[b, P+D .. P+tail) = 0 — the row is then the top tower's (padded) input — and the backward adds
 * g[b, P .. P+D) to demb[b, F-1, :]. */
int fx_dot_interact_fwd(const float* emb, int64_t emb_ld, int32_t F, int32_t D, int64_t B,
                        float* out, int64_t out_ld, int32_t tail, fx_stream_t stream);
int fx_dot_interact_bwd(const float* emb, int64_t emb_ld, const float* g, int64_t g_ld, int32_t tail,
                        int32_t F, int32_t D, int64_t B, float* demb, int64_t demb_ld,
                        fx_stream_t stream);
int fx_lr_fwd(const float* table1, const int32_t* ids, int64_t ids_ld,
              const int64_t* col_row_base, const int32_t* col_vocab, int32_t C,
              const float* dense, int64_t dense_ld, const float* num_w1, int32_t Fd,
              const float* bias, float* out, int64_t B, fx_scalars* scal,
              int64_t table1_ld /* row stride of table1 in floats; <= 0: 1 */, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * fp32 MFMA GEMM with fused epilogue (MLP_Block / CrossNetV2 / fc layers and their backward:
 * mlp_block.py:96, cross_net.py:126-129, autograd of aten::addmm).  Row-major:
 *     acc[M,N] = op(A)[M,K] . op(B)[K,N]
 *     transa = 0: A stored [M,K] (lda)   transa = 1: A stored [K,M]
 *     transb = 0: B stored [K,N] (ldb)   transb = 1: B stored [N,K]   (nn.Linear weight)
 * Uses v_mfma_f32_32x32x2_f32: exact fp32 products and fp32 accumulation (no TF32/bf16).
 * Epilogue, in this order (NULL members are skipped):
 *     z = acc + bias[n];  zout[m,n] = z;  act(z) (1 = relu);  z *= mul[m,n];
 *     z = mask[m,n] > 0 ? z : 0;  z += add[m,n];  C[m,n] = z
 * split_k > 1 splits K over blocks; partials go to workspace[split_k][M][N] and a second
 * (deterministic) kernel reduces them and applies the epilogue.  Skinny shapes (K <= 8; N <= 4
 * with transa = 0; M <= 4 with transa = 1, transb = 0 and a workspace) — the Linear(hidden -> 1)
 * head of every tower and its gradients — run on bandwidth-bound kernels instead of MFMA tiles.
 * ------------------------------------------------------------------------------------------ */
typedef struct fx_gemm_epilogue {
    const float* bias;
    float* zout;
    int64_t ldz;
    int32_t act;
    const float* mul;
    int64_t ldmul;
    const float* mask;
    int64_t ldmask;
    const float* add;
    int64_t ldadd;
    float* rowsum; /* optional extra output [M]: rowsum[m] = sum_k op(A)[m,k] — the bias gradient
                      when op(A) = dZ^T (fuses the column-sum pass into the dW GEMM); with
                      split_k > 1 the workspace needs split_k*M extra floats */
} fx_gemm_epilogue;

int fx_gemm_f32(int32_t transa, int32_t transb, int64_t M, int64_t N, int64_t K, const float* A,
                int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                const fx_gemm_epilogue* epi_host, int32_t split_k, float* workspace,
                fx_stream_t stream);

/* Several independent GEMMs in as few launches as possible (same semantics as calling fx_gemm_f32 on
 * each problem, in order; the outputs must not overlap each other or any input).  Up to four aligned,
 * non-skinny problems — the weight gradient dW = dZ^T X (transa = 1, transb = 0) and the input gradient
 * dX = dZ W (transa = 0, transb = 0) of one Linear / CrossNet layer, the two `aten::mm` of its autograd
 * (mlp_block.py:96, cross_net.py:128 at rank_model.py:320), which share dZ and are independent; for
 * DCNv2's parallel structure also the cross and the deep layer of one depth (DCNv2.py:108-132) — leave
 * as ONE grid on 128-row tiles with two workgroups per CU (launches that contain a K-split problem, i.e.
 * backward passes; two forward products go out as two launches): per problem a 128x128 tile, or 128x64
 * when that leaves fewer workgroups than CUs, and K slabs ~1024 deep.
 * A problem's split_k is the LARGEST number of K slabs its workspace holds (split_k * M * (N + 1)
 * floats); the library never uses more.  problems_host is a HOST array. */
typedef struct fx_gemm_problem {
    int32_t transa, transb;
    int64_t M, N, K;
    const float* A;
    int64_t lda;
    const float* B;
    int64_t ldb;
    float* C;
    int64_t ldc;
    const fx_gemm_epilogue* epilogue; /* host pointer or NULL */
    int32_t split_k;
    float* workspace;
} fx_gemm_problem;
int fx_gemm_f32_batch(const fx_gemm_problem* problems_host, int32_t n, fx_stream_t stream);

/* Column sums (bias gradients): out[n] = sum_m X[m,n].
 * Two-stage deterministic reduction; workspace >= FX_COLSUM_CHUNKS * N floats. */
#define FX_COLSUM_CHUNKS 64
int fx_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out, float* workspace,
              fx_stream_t stream);

/* ReLU backward for a tower whose LAST layer is activated (e.g. DCNv2 parallel_dnn,
 * mlp_block.py:80-81 with output_dim=None): out[i] = y[i] > 0 ? dy[i] : 0.  (Inner layers get
 * this mask for free in the dX GEMM epilogue.)  dy is [rows, cols] with row stride dy_ld (it may be a
 * column slice of the gradient of a torch.cat), y [rows, cols] with row stride y_ld (it may be a column
 * slice of the [cross | deep] buffer that feeds DCNv2's head); out is contiguous. */
int fx_mask_mul(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, float* out, int64_t rows,
                int64_t cols, fx_stream_t stream);

/* CrossNetV2 backward glue (autograd of cross_net.py:128), one pass over [n] elements:
 *     t[i]   = dxn[i] * x0[i]                         (gradient of W x_i + b)
 *     term   = dxn[i] * z[i] (+ dxn[i] if add_dxn)    (gradient reaching x_0 through the Hadamard)
 *     dx0[i] = init ? term : dx0[i] + term
 * dxn is [rows, cols] with row stride dxn_ld; the other operands are contiguous. */
int fx_cross_bwd_prep(const float* dxn, int64_t dxn_ld, const float* x0, const float* z, float* t,
                      float* dx0, int64_t rows, int64_t cols, int32_t init, int32_t add_dxn,
                      fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Output activation + loss, fused: p = sigmoid(logit); loss = mean BCE(p, y) with torch's
 * log clamp at -100 (BaseModel.get_output_activation rank_model.py:447-448 +
 * F.binary_cross_entropy torch_utils.py:95-98); dlogit = dloss/dlogit (torch's two-step
 * backward: (p-y)/max(p(1-p),1e-12)/B * p(1-p)).  prob/loss/dlogit may be NULL; y == NULL
 * computes the activation only (evaluate / predict).
 * ------------------------------------------------------------------------------------------ */
int fx_sigmoid_bce(const float* logit, const float* y, int64_t B, float* prob, float* loss,
                   float* dlogit, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The last mile of a binary-classification training step in one pass over the top hidden layer:
 * BaseModel.train_step (rank_model.py:307-323) runs forward -> compute_loss -> loss.backward(); for a
 * tower that ends in Linear(K -> 1) (MLP_Block's output layer, blocks/mlp_block.py:44-45; DCNv2's `fc`,
 * DCNv2.py:100) whose output IS the logit, these are
 *     logit[m]  = h[m, :] . w + bias (+ add[m])                       (the head's forward)
 *     loss      = mean_m BCE(sigmoid(logit[m]), y[m])                 (fx_sigmoid_bce's formulas)
 *     dlogit[m] = dloss/dlogit[m] * root_scale
 *     dz[m, k]  = dlogit[m] * w[k]   (zeroed where h[m, k] <= 0 for k >= mask_from: the ReLU below the head;
 *                                     mask_from < 0: no mask, 0: every column, 624: DCNv2's [cross | deep] input)
 *     dW[:]     = sum_m dlogit[m] h[m, :],   db = sum_m dlogit[m]
 * h: [M, K] row stride ldh; add: NULL or [M] with stride ldadd (DeepFM: the FM + first-order term);
 * dz: NULL or [M, K] row stride lddz; K % 4 == 0, 8 < K <= 2048, 16-byte aligned rows.  The logit is bit for
 * bit fx_gemm_f32's (transb, N = 1, bias + add epilogue), so training and evaluate see one function; the
 * sums over m are taken in a fixed order (deterministic).  workspace: fx_head_train_workspace(M, K) floats.
 * ------------------------------------------------------------------------------------------ */
int64_t fx_head_train_workspace(int64_t M, int64_t K);
int fx_head_train(const float* h, int64_t ldh, const float* w, const float* bias, const float* add,
                  int64_t ldadd, const float* y, int64_t M, int64_t K, int32_t mask_from, float root_scale,
                  float* logit, float* dlogit, float* dz, int64_t lddz, float* dW, float* db, float* loss,
                  float* workspace, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DIN target attention (fuxictr/pytorch/layers/attentions/target_attention.py:66-92) and Dice
 * (fuxictr/pytorch/layers/activations.py:24-51).  The attention MLP itself runs on fx_gemm_f32;
 * these are the pieces around it.  q: [B,E] (row stride q_ld); K: [B,L,E] addressed as
 * K[b*k_ldb + l*k_ldl + e] (so a strided view of the gather record works); ids: the raw id
 * columns of the sequence (mask = id != 0, DIN.py:125).
 *   fx_din_concat_fwd : X[b*L+l, :] = [q_b, k_bl, q_b - k_bl, q_b * k_bl]            ([B*L, 4E])
 *   fx_din_concat_bwd : dq[b,:], dK[b,l,:] from dX
 *   fx_din_pool_fwd   : out[b,:] = sum_l w[b,l] * (ids[b,l] != 0) * k_bl
 *   fx_din_pool_bwd   : dw[b,l], dK[b,l,:] from dout
 *   fx_dice_fwd       : y = p z + alpha (1-p) z, p = sigmoid(BN(z)); training != 0 uses (and
 *                       stores in stats[2H]) the batch mean / biased variance over ALL N rows and
 *                       updates the running statistics with `momentum` (unbiased variance), like
 *                       nn.BatchNorm1d(affine=False); training == 0 uses the running statistics
 *   fx_dice_bwd       : dz (through the batch statistics in training mode) and dalpha[H]
 * workspace: fx_dice_workspace_floats(H) floats.
 * ------------------------------------------------------------------------------------------ */
int fx_din_concat_fwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb, int64_t k_ldl,
                      int64_t B, int32_t L, int32_t E, float* out, fx_stream_t stream);
int fx_din_concat_bwd(const float* dx, const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                      int64_t k_ldl, int64_t B, int32_t L, int32_t E, float* dq, float* dK,
                      int64_t dk_ldb, int64_t dk_ldl, int32_t accumulate_dk, fx_stream_t stream);
int fx_din_pool_fwd(const float* w, const int32_t* ids, int64_t ids_ld, const float* K,
                    int64_t k_ldb, int64_t k_ldl, int64_t B, int32_t L, int32_t E, float* out,
                    fx_stream_t stream);
int fx_din_pool_bwd(const float* w, const int32_t* ids, int64_t ids_ld, const float* K,
                    int64_t k_ldb, int64_t k_ldl, const float* dout, int64_t B, int32_t L,
                    int32_t E, float* dw, float* dK, int64_t dk_ldb, int64_t dk_ldl,
                    fx_stream_t stream);
/* Row-sharded training (every rank holds a slice of the global batch): the two column reductions are
 * exposed on their own so that the host can all-reduce them and Dice normalises with the statistics
 * of the WHOLE batch, as the reference does on one device:
 *   fx_dice_local_sums     sums[0..H) = sum_rows z, sums[H..2H) = sum_rows z^2 of this rank's rows
 *   fx_dice_fwd_from_sums  mean / biased variance from (all-reduced) sums and n_total rows, running
 *                          statistics (momentum, unbiased), then the gate — fx_dice_fwd's apply pass
 *   fx_dice_bwd_local_sums sums3 = [dalpha | sum dzhat | sum dzhat*zhat] of this rank's rows
 *   fx_dice_bwd_from_sums  dz from (all-reduced) sums3[H..3H) and n_total (dalpha = sums3[0..H) of
 *                          each rank is summed with the other dense gradients)
 * (activations.py:40-51; new functionality: the reference is single-device) */
int64_t fx_dice_workspace_floats(int32_t H);
int fx_dice_local_sums(const float* Z, int64_t N, int32_t H, float* sums, float* workspace,
                       fx_stream_t stream);
int fx_dice_fwd_from_sums(const float* Z, int64_t N, int32_t H, const float* alpha, float eps,
                          float momentum, const float* sums, int64_t n_total, float* running_mean,
                          float* running_var, float* stats, float* Y, fx_stream_t stream);
int fx_dice_bwd_local_sums(const float* Z, const float* dY, int64_t N, int32_t H, const float* alpha,
                           float eps, const float* stats, float* sums3, float* workspace,
                           fx_stream_t stream);
int fx_dice_bwd_from_sums(const float* Z, const float* dY, int64_t N, int32_t H, const float* alpha,
                          float eps, const float* stats, const float* sums3, int64_t n_total,
                          float* dZ, fx_stream_t stream);
int fx_dice_fwd(const float* Z, int64_t N, int32_t H, const float* alpha, float eps, float momentum,
                int32_t training, float* running_mean, float* running_var, float* stats, float* Y,
                float* workspace, fx_stream_t stream);
int fx_dice_bwd(const float* Z, const float* dY, int64_t N, int32_t H, const float* alpha, float eps,
                int32_t training, const float* stats, float* dZ, float* dalpha, float* workspace,
                fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * DIN target attention with the attention MLP fused in
 * (fuxictr/pytorch/layers/attentions/target_attention.py:66-92 — concatenation [q, k, q-k, q*k],
 * MLP_Block(4E -> H, Dice, -> 1), mask, weighted sum over the sequence — and
 * fuxictr/pytorch/layers/activations.py:40-51, Dice; their autograd at rank_model.py:320).
 * Neither the [B*L, 4E] concatenation nor the [B*L, H] hidden tensor is written: every pass
 * recomputes the hidden layer on the matrix cores from q [B,E] and K [B,L,E] (addressed like
 * fx_din_concat_fwd).  Limits: E <= 16, H <= 64, one hidden layer.
 * W1: [H, 4E] row-major, b1: [H] or NULL, W2: [H], b2: [1] or NULL, stats: mean[H] | biased var[H],
 * mask: int32 [B, L] (row stride mask_ld; position kept when != 0; NULL = all kept).
 *   fx_din_attn_stats       sums[2H] = [sum h | sum h^2] over this rank's B*L positions
 *   fx_din_attn_stats       sums[2H] = [sum h | sum h^2] over the B*L positions (h = W1 x + b1); with
 *                           stats != NULL (single rank: no all-reduce in between) the statistics of
 *                           fx_dice_stats_from_sums(training) over n_total = B*L are finished in the same
 *                           launch (stats[2H], running statistics, num_batches_tracked)
 *   fx_dice_stats_from_sums training != 0: stats from (all-reduced) sums and n_total rows + running
 *                           statistics update (momentum, unbiased variance; num_batches_tracked += 1 when
 *                           the pointer is given), like nn.BatchNorm1d(affine=False); training == 0:
 *                           stats = running statistics
 *   fx_din_attn_fwd         a[b*L + l] = W2 . Dice(W1 x_bl + b1) + b2 (before the mask) and the
 *                           pooled output out[b,:] = sum_l a mask k_bl
 *   fx_din_attn_bwd_sums    from dout[B,E]: da[b*L + l] = mask (dout_b . k_bl) (written) and
 *                           sums5[5H] = [dalpha | sum dzhat | sum dzhat*zhat | dW2 | db2, 0...];
 *                           the host all-reduces [H, 3H) across ranks in sharded training
 *   fx_din_attn_bwd         dq[B,E], dK[B,L,E] (attention part + the pooling's share a mask dout),
 *                           dW1b1 = [dW1 (H*4E) | db1 (H)]
 * workspace: fx_din_attn_workspace_floats(B, L, E, H) floats.
 * ------------------------------------------------------------------------------------------ */
int64_t fx_din_attn_workspace_floats(int64_t B, int32_t L, int32_t E, int32_t H);
int fx_din_attn_stats(const float* q, int64_t q_ld, const float* K, int64_t k_ldb, int64_t k_ldl,
                      int64_t B, int32_t L, int32_t E, const float* W1, const float* b1, int32_t H,
                      float* sums, float* workspace, float* stats, float momentum, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, fx_stream_t stream);
int fx_dice_stats_from_sums(const float* sums, int32_t H, int64_t n_total, float momentum,
                            int32_t training, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, float* stats, fx_stream_t stream);
int fx_din_attn_fwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb, int64_t k_ldl,
                    int64_t B, int32_t L, int32_t E, const float* W1, const float* b1, int32_t H,
                    const float* alpha, float eps, const float* stats, const float* W2,
                    const float* b2, const int32_t* mask, int64_t mask_ld, float* a_out, float* out,
                    int64_t out_ld, fx_stream_t stream);
int fx_din_attn_bwd_sums(const float* q, int64_t q_ld, const float* K, int64_t k_ldb, int64_t k_ldl,
                         int64_t B, int32_t L, int32_t E, const float* W1, const float* b1,
                         int32_t H, const float* alpha, float eps, const float* stats,
                         const float* W2, const int32_t* mask, int64_t mask_ld, const float* dout,
                         int64_t dout_ld, float* da, float* sums5, float* workspace,
                         fx_stream_t stream);
int fx_din_attn_bwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb, int64_t k_ldl,
                    int64_t B, int32_t L, int32_t E, const float* W1, const float* b1, int32_t H,
                    const float* alpha, float eps, int32_t training, const float* stats,
                    const float* W2, const int32_t* mask, int64_t mask_ld, const float* a_logit,
                    const float* dout, int64_t dout_ld, const float* da, const float* sums5,
                    int64_t n_total, float* dq, int64_t dq_ld, int32_t dq_accumulate, float* dK,
                    int64_t dk_ldb, int64_t dk_ldl, float* dW1b1, float* workspace, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * xDeepFM Compressed Interaction Network layer, fused (compressed_interaction_net.py:54-76):
 *     Xn[b,o,d] = sum_{h,m} W[o, h*Mi+m] X0[b,h,d] Xi[b,m,d] + bias[o];  pool[b,o] = sum_d Xn[b,o,d]
 * X0: [B,F0,D] (sample stride x0_ld), Xi: [B,Mi,D] (xi_ld), W: [O, F0*Mi] (the Conv1d(k=1) weight),
 * Xn: [B,O,D] contiguous; pool (nullable) is written at pool[b*pool_ld + o], i.e. straight into its
 * slot of the concatenated pooling vector.  The einsum tensor [B, F0*Mi, D] is never materialised.
 * fx_cin_bwd: with g = dXn (nullable) + dpool (nullable, broadcast over d):
 *     dX0 (+)= ..., dXi = ..., partial[G][O*F0*Mi + O] = per-workgroup sums of dW and dbias
 *     (G = fx_cin_workgroups(), row stride partial_ld >= O*F0*Mi + O: the layers of a stack write column
 *     slices of ONE [G, sum] buffer; finish with one fx_colsum over G).
 * Limits: O*F0*Mi + O <= 30720 floats per call (LDS-resident weights), D <= 256.
 * D = 16, O <= 16, F0 <= 40, Mi <= 40 (the BASELINE xDeepFM: 39 fields, 16 dims, 16 maps) run on the
 * matrix cores: per sample the compress step is W [16 x F0*Mi] times the outer product [F0*Mi x 16],
 * formed in registers.  For those shapes fx_cin_wimg_floats > 0 and fx_cin_pack_w lays the W of up to 4
 * layers out in one launch, once per step, as the LDS images the forward and dX kernels copy (w_img,
 * 16-byte aligned, valid until W changes); w_img = NULL is accepted everywhere (the kernels then gather
 * the image from W themselves).
 * ------------------------------------------------------------------------------------------ */
int64_t fx_cin_workgroups(void);
int64_t fx_cin_wimg_floats(int32_t F0, int32_t Mi, int32_t D, int32_t O);
int fx_cin_pack_w(int32_t n_layers, const float* const* W, const int32_t* F0, const int32_t* Mi,
                  int32_t D, const int32_t* O, float* const* w_img, fx_stream_t stream);
int fx_cin_fwd(const float* X0, int64_t x0_ld, int32_t F0, const float* Xi, int64_t xi_ld,
               int32_t Mi, int32_t D, const float* W, const float* bias, int32_t O, float* Xn,
               float* pool, int64_t pool_ld, int64_t B, const float* w_img, fx_stream_t stream);
int fx_cin_bwd(const float* X0, int64_t x0_ld, int32_t F0, const float* Xi, int64_t xi_ld,
               int32_t Mi, int32_t D, const float* W, int32_t O, const float* dXn,
               const float* dpool, int64_t dpool_ld, float* dX0, int64_t dx0_ld,
               int32_t accumulate_dx0, float* dXi, int64_t dxi_ld, float* partial, int64_t partial_ld,
               int64_t B, const float* w_img, fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * On-device evaluation metrics for BaseModel.evaluate (rank_model.py:350-381, metrics.py:49-51):
 * binary logloss (sklearn.metrics.log_loss on float64: probabilities clipped to
 * [DBL_EPSILON, 1-DBL_EPSILON]) and AUC (sklearn.metrics.roc_auc_score = Mann-Whitney U with
 * average ranks for ties) over n predictions resident on the device.
 *   out_logloss_sum[0] = sum of the per-sample losses (divide by n)
 *   out_counts[0] = 2 * (rank sum of the positives, ranks 1-based, ties averaged)   (exact integer)
 *   out_counts[1] = number of positives
 *   AUC = (out_counts[0]/2 - n_pos (n_pos+1)/2) / (n_pos (n - n_pos))
 * 1 <= n <= 2^26; y_true in {0,1}; workspace from fx_binary_metrics_workspace_bytes(n).
 * ------------------------------------------------------------------------------------------ */
size_t fx_binary_metrics_workspace_bytes(int64_t n);
int fx_binary_metrics(const float* y_pred, const float* y_true, int64_t n, void* workspace,
                      size_t workspace_bytes, double* out_logloss_sum, uint64_t* out_counts,
                      fx_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused sparse front end / back end (csrc/fx_fused.hip).  For a FeatureEmbeddingDict whose id
 * columns own disjoint tables laid out in column order (every categorical schema of the BASELINE
 * configs) the per-step embedding work collapses to a handful of launches; the entry points above
 * stay as the general path (sequence columns, row-sharded tables, B > 8192).
 *
 * fx_row_state: one packed table with its optimizer state; several of them may share one id plan
 * (the D=16 tables of FeatureEmbedding and the D=1 tables LogisticRegression builds at
 * logistic_regression.py:44 have identical row layouts).
 *
 * fx_dedup_catchup = fx_dedup (column fast path) + fx_opt_begin_step + fx_adam_catchup, 2 launches:
 *   launch 1 sorts every id column in LDS and, when begin_scal != NULL, opens the optimizer step
 *   (step += 1, Adam bias corrections: torch.optim.Adam as stepped at rank_model.py:322);
 *   launch 2 derives uniq_row / seg_start / n_unique / sorted_uid (as fx_dedup) and replays the
 *   missed zero-gradient Adam steps of every unique row up to step + upto_offset in each of the
 *   n_tables (<= 4) table groups, so the lookup that follows reads the rows dense Adam would hold.
 *   n_tables = 0: de-dup only.  workspace: >= 2 * round_up(4*B*C, 256) bytes.  B <= 8192, C <= 256.
 *
 * bf16 table storage (BASELINE north_star "vectorised fp32/bf16 gathers"; opt-in `emb_dtype: bf16`):
 *   table_dtype = FX_BF16 makes `table` a bf16 [rows, D] array — a row of D = 16 is 32 bytes, read by
 *   4 lanes as 8-byte loads and widened; every sum (FM, first-order term), the Adam moments and the
 *   update arithmetic stay fp32; updated rows are rounded to nearest-even bf16.  The D=1 table of
 *   the first-order term is always fp32.
 *
 * fx_emb_fm_fwd = fx_emb_gather_fwd + fx_lr_fwd + fx_fm_fwd in ONE launch, one wave per sample:
 *   out     the [B, F, D] record (feature_embedding.py:261-297, :230-259), written once
 *   lr_out  [B]  sum_c table1[col_row_base[c] + ids[b,c]] + sum_j dense[b,j] num_w1[j] + bias1
 *                (LogisticRegression.forward, logistic_regression.py:46-59); NULL = not wanted
 *   fm_out  [B]  0.5 * sum_d ((sum_f e_fd)^2 - sum_f e_fd^2) over all C + Fd fields of the record
 *                (InnerProductInteraction "product_sum", inner_product.py:55-62); NULL = not wanted
 *   fm_lr_out [B] fm_out + lr_out (FactorizationMachine.forward, factorization_machine.py:46-59)
 *   S       [B, D] the per-dimension field sums, kept for fx_emb_fm_bwd; NULL = not wanted
 *
 * fx_emb_fm_bwd: autograd of the above, 2 launches.  Per unique row u (runs of sorted_pos given by
 *   seg_start / sorted_uid, as produced by fx_dedup_catchup with sorted_uid != NULL):
 *     G[u,:]  = sum over the row's lookups (b,c) of
 *               drec[b, col_out_off[c] + :] + g_fm[b] * (S[b,:] - rec[b, col_out_off[c] + :])
 *     G1[u]   = sum over the row's lookups of g_lr[b]                       (the D=1 table's row)
 *   and the fixed-order partial sums of ||G||^2 / ||G1||^2 (fx_emb_fm_bwd_partials(n_max, D) entries
 *   each) for clip_grad_norm_ (rank_model.py:321).  drec (gradient of the record from the layers
 *   that read it), g_fm, g_lr may each be NULL (that term is absent).  The work is handed out by
 *   SORTED LOOKUP, 256 per workgroup, not by row: the hot row of a 3-row table (2800 of 4096
 *   lookups) costs what 2800 cold rows cost; a run cut by a workgroup boundary is finished by
 *   launch 2 in workgroup order, so the result is deterministic.  n_max must be B*C.
 *   Numeric features: dnum_w[j,:] = sum_b dense[b,j] * (value of slot num_out_off[j]), dnum_w1[j] =
 *   sum_b dense[b,j] g_lr[b], dbias1 = sum_b g_lr[b]  (autograd of feature_embedding.py:280-282
 *   and logistic_regression.py:55-58), partial sums in launch 1, finals in launch 2.
 *   workspace: fx_emb_fm_bwd_workspace_floats(n_max, D, Fd) floats.
 *
 * fx_sparse_adam_multi / fx_sparse_sgd_multi: fx_sparse_adam / fx_sparse_sgd for every table group
 *   of one de-dup result in one launch (tables[t].G = that group's reduced gradient).
 *
 * fx_adam_catchup_all: fx_adam_catchup over EVERY row of one table (uniq_row = NULL there), for fp32
 *   or bf16 tables: the flush of the exact mode before evaluate / save / a learning-rate change.
 *
 * fx_adam_catchup_rows: fx_adam_catchup over the unique rows of ANY de-dup result, for fp32 or bf16
 *   tables and for every table group that shares the id plan (<= 4) in ONE launch: the catch-up of the
 *   generic de-dup path (sequence columns that alias a table, batches beyond fx_dedup_catchup's
 *   limits).  Replaces what dense torch.optim.Adam does to untouched rows (torch_utils.py:76,
 *   rank_model.py:322), like fx_adam_catchup.
 *
 * fx_pack_columns_multi: fx_pack_columns with one destination per column (outs_host[c] = address of
 *   out[0, first column], out_lds_host[c] its row stride, out_dtypes_host[c] FX_I32 | FX_F32), so the
 *   id block, the numeric block and the label of a batch (rank_model.py:169-204, feature_embedding.py
 *   :280-291) are cast in ONE launch.  <= 96 columns per call.
 * ------------------------------------------------------------------------------------------ */
typedef struct fx_row_state {
    void* table;        /* [rows, D], fp32 or bf16 (table_dtype) */
    float* m;           /* Adam moments (NULL for SGD) */
    float* v;
    int32_t* last_step; /* [rows] step of the row's last update (NULL: not tracked) */
    const float* G;     /* [n_max, D] reduced gradient, update entry points only */
    int32_t D;
    int32_t table_dtype; /* FX_F32 | FX_BF16: storage of `table` only; m, v, G are fp32 */
    /* row strides in elements of the respective array (0: packed = D, D, D, 1).  Round 6: the "row record" —
     * one [p | m | v | last_step] record of W floats per row, `table`, `m`, `v`, `last_step` pointing at its
     * four fields with all four strides = W — makes a row's catch-up / update ONE scattered access instead of
     * four (scripts/ubench/row_record.hip: 13.0 -> 5.9 us for 25 K rows out of 33.76 M). */
    int64_t table_ld, m_ld, v_ld, last_ld;
} fx_row_state;

int fx_dedup_catchup(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                     const int64_t* col_row_base, const int32_t* col_vocab, const int32_t* col_pad,
                     void* workspace, size_t workspace_bytes, uint32_t* sorted_key,
                     uint32_t* sorted_pos, uint32_t* uniq_row, uint32_t* seg_start,
                     int32_t* n_unique, uint32_t* sorted_uid /* or NULL */,
                     fx_scalars* begin_scal /* or NULL */, const fx_row_state* tables_host,
                     int32_t n_tables, int32_t upto_offset, const fx_scalars* scal,
                     fx_stream_t stream);
int fx_emb_fm_fwd(const void* table, int32_t table_dtype, int32_t D, const int32_t* ids, int64_t ids_ld,
                  const int64_t* col_row_base, const int32_t* col_vocab, const int64_t* col_out_off,
                  int32_t C, const float* dense, int64_t dense_ld, const float* num_w,
                  const int64_t* num_out_off, int32_t Fd, float* out, int64_t out_ld, int64_t B,
                  const float* table1, const float* num_w1, const float* bias1, float* lr_out,
                  float* fm_out, float* fm_lr_out, float* S, fx_scalars* scal,
                  int64_t table_ld /* row stride of `table` in elements; <= 0: D */,
                  int64_t table1_ld /* row stride of `table1`; <= 0: 1 */,
                  int64_t zero_off0, int32_t zero_n0, int64_t zero_off1, int32_t zero_n1
                  /* two ranges of reserved floats per record row (offset, count; count 0 = none) that the
                     launch clears: slots a later kernel of the step fills (DIN's attended vector, DLRM's
                     bottom-tower vector) */,
                  fx_stream_t stream);
/* fx_owner_fetch_rows: owner side of the row-sharded forward for every table group of one exchange
 *   (<= 4) in ONE launch.  For each unique owned row u of the owner-side de-dup (uniq_row / seg_start /
 *   sorted_pos / n_unique of fx_dedup_sorted_runs over the n_total received ids): catchup != 0 replays
 *   the row's missed zero-gradient Adam steps up to step + upto_offset exactly as fx_adam_catchup_rows
 *   (tables need m, v, last_step), then the row is written into send[sorted_pos[i], off_t .. off_t + D_t)
 *   for every entry i of its run (one per requesting rank); entries that asked for nothing (pad ids)
 *   and the pad columns of the [n_total, ld] block are zeroed; zero_row / zero_w: a row of zero_w
 *   floats cleared in the same launch (the pad row of the block the received rows will land in; NULL /
 *   0 = none).  Replaces, on the owning rank, the
 *   row reads of aten::embedding (feature_embedding.py:283-291) and dense Adam's step on untouched
 *   rows (torch_utils.py:76).  fp32 or bf16 tables (table_dtype): a bf16 row travels widened to fp32, after the
 *   rounding its stored copy went through. */
int fx_owner_fetch_rows(const fx_row_state* tables_host, const int32_t* off_host, int32_t n_tables,
                        const uint32_t* uniq_row, const uint32_t* seg_start,
                        const uint32_t* sorted_pos, const int32_t* n_unique, int64_t n_total,
                        float* send, int64_t ld, int32_t catchup, int32_t upto_offset,
                        const fx_scalars* scal, float* zero_row, int32_t zero_w, fx_stream_t stream);
int64_t fx_emb_fm_bwd_partials(int64_t n_lookups, int32_t D);
int64_t fx_emb_fm_bwd_workspace_floats(int64_t n_lookups, int32_t D, int32_t Fd);
int fx_emb_fm_bwd(const float* drec, int64_t drec_ld, const float* rec, int64_t rec_ld,
                  const float* S, const float* g_fm, const float* g_lr, const int64_t* col_out_off,
                  int32_t C, int32_t D, const uint32_t* sorted_pos, const uint32_t* sorted_uid,
                  const uint32_t* seg_start, const int32_t* n_unique, int64_t n_max, float* G,
                  float* sq_partials, float* G1, float* sq1_partials, const float* dense,
                  int64_t dense_ld, const int64_t* num_out_off, int32_t Fd, int64_t B,
                  float* dnum_w, float* dnum_w1, float* dbias1, float* workspace,
                  fx_stream_t stream);
int fx_sparse_adam_multi(const fx_row_state* tables_host, int32_t n_tables, const uint32_t* uniq_row,
                         const int32_t* n_unique, int64_t n_max, const fx_scalars* scal,
                         fx_stream_t stream);
int fx_sparse_sgd_multi(const fx_row_state* tables_host, int32_t n_tables, const uint32_t* uniq_row,
                        const int32_t* n_unique, int64_t n_max, const fx_scalars* scal,
                        fx_stream_t stream);
int fx_adam_catchup_all(const fx_row_state* table_host, int64_t total_rows, int32_t upto_offset,
                        const fx_scalars* scal, fx_stream_t stream);
int fx_adam_catchup_rows(const fx_row_state* tables_host, int32_t n_tables, const uint32_t* uniq_row,
                         const int32_t* n_unique, int64_t n_max, int32_t upto_offset,
                         const fx_scalars* scal, fx_stream_t stream);
int fx_pack_columns_multi(const void* const* cols_host, const int32_t* dtypes_host,
                          const int32_t* widths_host, void* const* outs_host,
                          const int32_t* out_dtypes_host, const int64_t* out_lds_host, int32_t ncols,
                          int64_t B, fx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FXCTR_H */
