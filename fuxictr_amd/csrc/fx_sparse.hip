// fx_sparse.hip — sparse side of the training step: index de-duplication, run-wise gradient
// reduction to unique rows, global-norm clip coefficient, sparse-row Adam / SGD with the
// "exact" catch-up replay, and the multi-tensor dense optimizer kernels.
//
// What this replaces in the reference (paths relative to the reference checkout):
//   aten::embedding_dense_backward (zero-filled [V,D] grad + index_add), triggered at
//     fuxictr/pytorch/models/rank_model.py:320
//   nn.utils.clip_grad_norm_ over every parameter            rank_model.py:321
//   torch.optim.Adam / SGD step over every table row          rank_model.py:322, torch_utils.py:76
// None of the table-sized passes exist here: the gradient only ever exists for the unique rows
// a batch touches.  All reductions use a fixed order (stable sort by row, ascending lookup
// position inside a run, fixed trees) so results are run-to-run deterministic.
#include "fx_common.h"
#include <stdlib.h>


// ---------------------------------------------------------------------------------------------
// de-duplication
// ---------------------------------------------------------------------------------------------
// key = global packed row g, or — with the table row-sharded over n_shards ranks — the
// owner-major pair (g % n_shards) * rows_per_shard + g / n_shards, so that a sort groups the
// lookups by owning rank and the key itself carries (owner, local row).
__global__ void k_opt_begin_step(fx_scalars* sc);

__global__ void k_zero_words(int32_t* p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0;
}

__global__ __launch_bounds__(256) void k_build_keys(const int32_t* ids, int64_t ids_ld, int64_t n,
                                                    int C, const int64_t* col_row_base,
                                                    const int32_t* col_vocab,
                                                    const int32_t* col_pad, uint32_t sentinel,
                                                    uint32_t n_shards, uint32_t rows_per_shard,
                                                    uint32_t* keys, uint32_t* zero, int zero_words,
                                                    fx_scalars* begin_scal) {
    // (also opens the optimizer step when asked to — fx_opt_begin_step fused — and clears the sort's
    // first counter set, fx_sort_zero_words: two launches saved)
    if (begin_scal != nullptr && blockIdx.x == 0 && threadIdx.x == 0) fx_begin_step_dev(begin_scal);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < zero_words; i += gridDim.x * blockDim.x)
        zero[i] = 0u;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / C;
        const int c = (int)(i - b * C);
        const int32_t id = ids[b * ids_ld + c];
        uint32_t key = sentinel;
        if (id >= 0 && id < col_vocab[c] && id != col_pad[c]) {
            const uint32_t g = (uint32_t)(col_row_base[c] + id);
            key = n_shards > 1 ? (g % n_shards) * rows_per_shard + g / n_shards : g;
        }
        keys[i] = key;
    }
}

__global__ __launch_bounds__(256) void k_keys_one_table(const int32_t* ids, int64_t n, int32_t vocab,
                                                        int32_t pad, uint32_t sentinel,
                                                        uint32_t* keys) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t id = ids[i];
        keys[i] = (id >= 0 && id < vocab && id != pad) ? (uint32_t)id : sentinel;
    }
}

// Fast path when every id column owns a distinct table and the tables are laid out in column
// order: keys of column c all lie in [base_c, base_c + V_c), so sorting each column on its own
// yields the globally sorted array (segment c = [c*B, (c+1)*B)).  Padding / bad-id lookups keep a
// key inside their column (so the array stays sorted) but carry pos = 0xFFFFFFFF = "contributes
// nothing".  The two launches are fx_fused.hip's (k_sort_columns3: one hand-written in-LDS radix
// sort per column; k_finish_catchup without tables: scan + scatter of the unique rows).

// ---------------------------------------------------------------------------------------------
// Unique rows of a SORTED key array (generic path, owner-side merge): two launches, no library.
//   k_head_blocks           heads (first lookup of a row) per 1024-item tile -> block_sum[tile]
//   k_scan_scatter_unique   tile offset = sum of the tiles before it (<= a few hundred words), an
//                           in-tile scan of the head flags, then uniq_row / seg_start / sorted_uid /
//                           n_unique exactly as the library-scan version of round 2 wrote them.
// (Round 2: rocprim::inclusive_scan — an init launch + a decoupled-look-back scan, 17 us for DIN's
// 209 K lookups — followed by a scatter launch.)
// ---------------------------------------------------------------------------------------------
#define FX_HS_TILE 1024

__device__ __forceinline__ uint32_t fx_head_flag(const uint32_t* key, int64_t i, int64_t n,
                                                 uint32_t sentinel) {
    if (i >= n) return 0u;
    const uint32_t k = key[i];
    return (k != sentinel && (i == 0 || key[i - 1] != k)) ? 1u : 0u;
}

__global__ __launch_bounds__(256) void k_head_blocks(const uint32_t* key, int64_t n, uint32_t sentinel,
                                                     uint32_t* block_sum) {
    __shared__ uint32_t red[4];
    const int64_t base = (int64_t)blockIdx.x * FX_HS_TILE + threadIdx.x * 4;
    uint32_t h = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) h += fx_head_flag(key, base + j, n, sentinel);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor(h, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = h;
    __syncthreads();
    if (threadIdx.x == 0) block_sum[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_scan_scatter_unique(const uint32_t* key, int64_t n,
                                                             uint32_t sentinel,
                                                             const uint32_t* block_sum,
                                                             uint32_t* uniq_row, uint32_t* seg_start,
                                                             int32_t* n_unique, uint32_t* sorted_uid) {
    __shared__ uint32_t red[4], wsum[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // heads in the tiles before this one
    uint32_t off = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) off += block_sum[b];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) off += __shfl_xor(off, o, 64);
    if (lane == 0) red[w] = off;
    const int64_t base = (int64_t)blockIdx.x * FX_HS_TILE + threadIdx.x * 4;
    uint32_t f[4], h = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[j] = fx_head_flag(key, base + j, n, sentinel);
        h += f[j];
    }
    uint32_t inc = h;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t y = __shfl_up(inc, o, 64);
        if (lane >= o) inc += y;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    uint32_t u = (red[0] + red[1]) + (red[2] + red[3]) + inc - h;
    for (int ww = 0; ww < w; ++ww) u += wsum[ww];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t i = base + j;
        if (i >= n) break;
        u += f[j];                                  // inclusive count of heads up to i
        const uint32_t k = key[i];
        if (k == sentinel) {
            if (sorted_uid) sorted_uid[i] = 0xFFFFFFFFu;
            if (i == 0) {                           // no valid lookup at all
                *n_unique = 0;
                seg_start[0] = 0;
            }
            continue;
        }
        if (sorted_uid) sorted_uid[i] = u - 1;
        if (f[j]) {
            uniq_row[u - 1] = k;
            seg_start[u - 1] = (uint32_t)i;
        }
        if (i == n - 1 || key[i + 1] == sentinel) {  // last valid lookup
            seg_start[u] = (uint32_t)(i + 1);
            *n_unique = (int32_t)u;
        }
    }
}

static int fx_unique_from_sorted(const uint32_t* sorted_key, int64_t n, uint32_t sentinel,
                                 uint32_t* block_sum, uint32_t* uniq_row, uint32_t* seg_start,
                                 int32_t* n_unique, uint32_t* sorted_uid, hipStream_t s) {
    const unsigned nblk = (unsigned)fx_ceil_div(n, FX_HS_TILE);
    hipLaunchKernelGGL(k_head_blocks, dim3(nblk), dim3(256), 0, s, sorted_key, n, sentinel, block_sum);
    hipLaunchKernelGGL(k_scan_scatter_unique, dim3(nblk), dim3(256), 0, s, sorted_key, n, sentinel,
                       block_sum, uniq_row, seg_start, n_unique, sorted_uid);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

static inline size_t fx_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

static hipError_t fx_dedup_temp_bytes(int64_t n, size_t* sort_bytes, size_t* scan_bytes) {
    *sort_bytes = fx_sort_temp_bytes(n);       // the device-wide sort is fx_sort.hip's
    *scan_bytes = (size_t)fx_ceil_div(n, FX_HS_TILE) * sizeof(uint32_t) + 256;   // k_head_blocks
    return hipSuccess;
}

extern "C" size_t fx_dedup_workspace_bytes(int64_t n_lookups) {
    if (n_lookups <= 0) return 256;
    size_t sort_bytes = 0, scan_bytes = 0;
    if (fx_dedup_temp_bytes(n_lookups, &sort_bytes, &scan_bytes) != hipSuccess) {
        fx_set_error("fx_dedup_workspace_bytes: size query failed");
        return 0;
    }
    const size_t arr = fx_align_up((size_t)n_lookups * sizeof(uint32_t), 256);
    const size_t tmp = fx_align_up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes, 256);
    return 3 * arr + tmp + 256;
}

extern "C" int fx_dedup(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                        const int64_t* col_row_base, const int32_t* col_vocab,
                        const int32_t* col_pad, int64_t total_rows, void* workspace,
                        size_t workspace_bytes, uint32_t* sorted_key, uint32_t* sorted_pos,
                        uint32_t* uniq_row, uint32_t* seg_start, int32_t* n_unique,
                        uint32_t* sorted_uid, int32_t n_shards, int32_t columns_sorted,
                        fx_scalars* begin_scal, fx_stream_t stream) {
    FX_CHECK_ARG(B >= 0 && C >= 0, "fx_dedup: negative size");
    FX_CHECK_ARG(total_rows > 0 && total_rows < (int64_t)0xFFFFFFFFLL,
                 "fx_dedup: total_rows=%lld must be in (0, 2^32-1)", (long long)total_rows);
    FX_CHECK_ARG(n_shards >= 1, "fx_dedup: n_shards must be >= 1");
    const int64_t rows_per_shard = fx_ceil_div(total_rows, n_shards);
    FX_CHECK_ARG(rows_per_shard * n_shards < (int64_t)0xFFFFFFFFLL,
                 "fx_dedup: sharded key space exceeds 32 bits");
    FX_CHECK_ARG(n_unique && seg_start, "fx_dedup: null output");
    hipStream_t s = fx_hip_stream(stream);
    const int64_t n = B * (int64_t)C;
    if (n == 0) {
        if (begin_scal) hipLaunchKernelGGL(k_opt_begin_step, dim3(1), dim3(64), 0, s, begin_scal);
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, n_unique, 1);
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, reinterpret_cast<int32_t*>(seg_start), 1);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    FX_CHECK_ARG(n < (int64_t)0x7FFFFFFF, "fx_dedup: too many lookups (%lld)", (long long)n);
    FX_CHECK_ARG(ids && col_row_base && col_vocab && col_pad && workspace && sorted_key &&
                     sorted_pos && uniq_row,
                 "fx_dedup: null pointer");
    size_t sort_bytes = 0, scan_bytes = 0;
    FX_CHECK_HIP(fx_dedup_temp_bytes(n, &sort_bytes, &scan_bytes));
    const size_t arr = fx_align_up((size_t)n * sizeof(uint32_t), 256);
    size_t tmp = fx_align_up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes, 256);
    FX_CHECK_ARG(workspace_bytes >= 3 * arr + tmp, "fx_dedup: workspace too small (%zu < %zu)",
                 workspace_bytes, 3 * arr + tmp);
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(w);
    uint32_t* keys_tmp = reinterpret_cast<uint32_t*>(w + arr);
    uint32_t* scan = reinterpret_cast<uint32_t*>(w + 2 * arr);   // the sort's value scratch first
    void* temp = w + 3 * arr;
    const uint32_t sentinel = (uint32_t)(n_shards > 1 ? rows_per_shard * n_shards : total_rows);

    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 4096) blocks = 4096;
    if (columns_sorted == 1 && n_shards == 1 && B <= 8192 && C <= 256) {
        // (measured in round 1: rocprim's segmented_radix_sort takes 81 us for 26 x 4096, its device
        // merge sort 57 us; one in-LDS workgroup sort per column over only the needed bits is the path)
        return fx_dedup_columns_launch(ids, ids_ld, B, C, col_row_base, col_vocab, col_pad, keys_in, scan,
                                       sorted_key, sorted_pos, uniq_row, seg_start, n_unique,
                                       sorted_uid, begin_scal, s);
    } else if (columns_sorted == 2 && n_shards == 1 && fx_dedup_buckets_ok(n) &&
               fx_dedup_buckets_bytes(n) <= workspace_bytes) {
        // round 6: sequence columns that alias a table, shared tables, B > 8192 — rows hashed into 256 buckets by
        // their low 8 bits, each bucket sorted by one workgroup in LDS (fx_dedup_lds.hip): 4 launches instead of
        // 10; unique rows come out grouped by (row & 255, row >> 8) instead of ascending
        return fx_dedup_buckets_launch(ids, ids_ld, B, C, col_row_base, col_vocab, col_pad, sentinel, workspace,
                                       sorted_key, sorted_pos, uniq_row, seg_start, n_unique, sorted_uid,
                                       begin_scal, s);
    } else {
        hipLaunchKernelGGL(k_build_keys, dim3((unsigned)blocks), dim3(256), 0, s, ids, ids_ld, n,
                           (int)C, col_row_base, col_vocab, col_pad, sentinel, (uint32_t)n_shards,
                           (uint32_t)rows_per_shard, keys_in, reinterpret_cast<uint32_t*>(temp),
                           (int)fx_sort_zero_words(n), begin_scal);
        FX_CHECK_LAUNCH();
        // (key, lookup index) pairs, over only the bits the key space needs (keys <= sentinel):
        // c4's 2.8 M rows take 22 of 32 bits = 3 radix passes
        unsigned end_bit = 1;
        while (end_bit < 32 && (sentinel >> end_bit) != 0u) ++end_bit;
        const int rc = fx_sort_pairs_u32(keys_in, nullptr, sorted_key, sorted_pos, keys_tmp, scan, n,
                                         end_bit, temp, true, s);
        if (rc != FX_OK) return rc;
    }
    return fx_unique_from_sorted(sorted_key, n, sentinel, reinterpret_cast<uint32_t*>(temp), uniq_row,
                                 seg_start, n_unique, sorted_uid, s);
}

// ---------------------------------------------------------------------------------------------
// fx_dedup_sorted_runs: the keys are R consecutive runs, each already ascending (what the owner of
// a row-sharded table receives: every peer's unique rows in ascending order, pad rows at the tail).
// A stable merge needs no sort: the output rank of element j of run r is
//     j + sum_{r' < r} upper_bound(run r', key) + sum_{r' > r} lower_bound(run r', key)
// — R-1 binary searches per element in ONE launch (the generic path's rocPRIM merge sort takes a
// block sort + ~8 merge passes, ~55-75 us for 160 K keys; this is bound by L2 latency instead).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fx_run_key(const int32_t* __restrict__ ids, int64_t i,
                                                int32_t vocab, int32_t pad) {
    const int32_t id = ids[i];
    return (id >= 0 && id < vocab && id != pad) ? (uint32_t)id : (uint32_t)vocab;
}

// (keys are formed on the fly from the ids — the sentinel `vocab` for pad / out-of-range entries: the
// separate key-building launch of round 2 is gone)
__global__ __launch_bounds__(256) void k_merge_runs(const int32_t* __restrict__ ids, int32_t vocab,
                                                    int32_t pad, int n_runs, int run_len,
                                                    uint32_t* __restrict__ sorted_key,
                                                    uint32_t* __restrict__ sorted_pos) {
    const int64_t n = (int64_t)n_runs * run_len;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int r = (int)(i / run_len);
        const uint32_t k = fx_run_key(ids, i, vocab, pad);
        uint32_t rank = (uint32_t)(i - (int64_t)r * run_len);
        for (int q = 0; q < n_runs; ++q) {
            if (q == r) continue;
            const int64_t run0 = (int64_t)q * run_len;
            int lo = 0, hi = run_len;
            if (q < r) {                       // elements <= k come first (stable: earlier run wins)
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (fx_run_key(ids, run0 + mid, vocab, pad) <= k) lo = mid + 1; else hi = mid;
                }
            } else {                           // elements < k come first
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (fx_run_key(ids, run0 + mid, vocab, pad) < k) lo = mid + 1; else hi = mid;
                }
            }
            rank += (uint32_t)lo;
        }
        sorted_key[rank] = k;
        sorted_pos[rank] = (uint32_t)i;
    }
}

extern "C" int fx_dedup_sorted_runs(const int32_t* ids, int32_t n_runs, int64_t run_len,
                                    int32_t vocab, int32_t pad, void* workspace,
                                    size_t workspace_bytes, uint32_t* sorted_key,
                                    uint32_t* sorted_pos, uint32_t* uniq_row, uint32_t* seg_start,
                                    int32_t* n_unique, uint32_t* sorted_uid, fx_stream_t stream) {
    FX_CHECK_ARG(n_runs >= 1 && n_runs <= 4096 && run_len >= 0, "fx_dedup_sorted_runs: bad shape");
    FX_CHECK_ARG(vocab > 0, "fx_dedup_sorted_runs: vocab must be > 0");
    FX_CHECK_ARG(n_unique && seg_start, "fx_dedup_sorted_runs: null output");
    hipStream_t s = fx_hip_stream(stream);
    const int64_t n = (int64_t)n_runs * run_len;
    if (n == 0) {
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, n_unique, 1);
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, reinterpret_cast<int32_t*>(seg_start), 1);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    FX_CHECK_ARG(n < (int64_t)0x7FFFFFFF, "fx_dedup_sorted_runs: too many keys (%lld)", (long long)n);
    FX_CHECK_ARG(ids && workspace && sorted_key && sorted_pos && uniq_row,
                 "fx_dedup_sorted_runs: null pointer");
    size_t sort_bytes = 0, scan_bytes = 0;
    FX_CHECK_HIP(fx_dedup_temp_bytes(n, &sort_bytes, &scan_bytes));
    const size_t arr = fx_align_up((size_t)n * sizeof(uint32_t), 256);
    size_t tmp = fx_align_up(sort_bytes > scan_bytes ? sort_bytes : scan_bytes, 256);
    FX_CHECK_ARG(workspace_bytes >= 3 * arr + tmp,
                 "fx_dedup_sorted_runs: workspace too small (%zu < %zu)", workspace_bytes,
                 3 * arr + tmp);
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(w);
    uint32_t* scan = reinterpret_cast<uint32_t*>(w + 2 * arr);
    void* temp = w + 3 * arr;
    const uint32_t sentinel = (uint32_t)vocab;       // = total_rows of this one-table key space
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 4096) blocks = 4096;
    // keys: the id itself, sentinel for pad / out-of-range ids (each run stays ascending: its pad
    // entries sit at the tail and map to the largest key)
    hipLaunchKernelGGL(k_merge_runs, dim3((unsigned)blocks), dim3(256), 0, s, ids, vocab, pad,
                       (int)n_runs, (int)run_len, sorted_key, sorted_pos);
    FX_CHECK_LAUNCH();
    (void)scan;
    (void)keys_in;
    return fx_unique_from_sorted(sorted_key, n, sentinel, reinterpret_cast<uint32_t*>(temp), uniq_row,
                                 seg_start, n_unique, sorted_uid, s);
}

// ---------------------------------------------------------------------------------------------
// gradient reduction to unique rows, three launches:
//   short : a lane group (D/VEC lanes) owns one unique row and sums its run when it has at most
//           FX_LONG_RUN lookups; longer runs (hot rows of tiny tables: a 3-row table at B=4096
//           has runs of ~1365) are appended to a list (order of the list is irrelevant to results)
//   long  : one workgroup per listed row, group g sums lookups g, g+RPB, ... and a fixed LDS
//           tree combines the groups, so a hot row costs ~run/RPB load rounds instead of `run`
//   sqnorm: block b reduces ||G||^2 of rows [b*RPB, (b+1)*RPB) in a fixed order -> partial[b]
// ---------------------------------------------------------------------------------------------
#define FX_LONG_RUN 32

struct ReduceArgs {
    const float* dout;
    int64_t dout_ld;
    const int64_t* col_out_off;
    const uint32_t* sorted_pos;
    const uint32_t* seg_start;
    const int32_t* n_unique;
    float* G;
    float* sq_partials;
    int32_t* scratch;   // [0] = number of long rows, [1..] = their unique-row indices
    int32_t C, D, lanes_log2;
    const int32_t* col_denom;   // SCALED kernels only: per id column, index into denom or -1
    const float* denom;
    int64_t denom_ld;
};

// value of lookup position p = b*C + c (SCALED: divided by its sample's pooling denominator)
template <int VEC, bool SCALED>
__device__ __forceinline__ void fx_load_lookup(const ReduceArgs& a, uint32_t p, int d0,
                                               float (&v)[VEC]) {
    const uint32_t b = p / (uint32_t)a.C, c = p - b * (uint32_t)a.C;
    fx_load<VEC>(a.dout + (int64_t)b * a.dout_ld + a.col_out_off[c] + d0, v);
    if constexpr (SCALED) {
        const int32_t j = a.col_denom[c];
        if (j >= 0) {
            const float den = a.denom[(int64_t)b * a.denom_ld + j];
#pragma unroll
            for (int k = 0; k < VEC; ++k) v[k] = v[k] / den;
        }
    }
}

template <int VEC, bool SCALED>
__device__ __forceinline__ void fx_accum_lookup(const ReduceArgs& a, uint32_t i, int d0,
                                                float (&acc)[VEC]) {
    const uint32_t p = a.sorted_pos[i];
    if (p == 0xFFFFFFFFu) return;   // padding_idx / bad-id lookup: contributes nothing
    float v[VEC];
    fx_load_lookup<VEC, SCALED>(a, p, d0, v);
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] += v[k];
}

// N lookups in flight per lane (positions first, then the N independent row loads), summed in
// ascending order: same result as the one-at-a-time loop with ~N x fewer dependent latencies.
template <int VEC, bool SCALED, int N>
__device__ __forceinline__ uint32_t fx_accum_chunks(const ReduceArgs& a, uint32_t i, uint32_t end,
                                                    uint32_t stride, int d0, float (&acc)[VEC]) {
    for (; i + (N - 1) * stride < end; i += N * stride) {
        uint32_t p[N];
#pragma unroll
        for (int j = 0; j < N; ++j) p[j] = a.sorted_pos[i + j * stride];
        float v[N][VEC];
#pragma unroll
        for (int j = 0; j < N; ++j) {
            if (p[j] == 0xFFFFFFFFu) {   // padding_idx / bad-id lookup: contributes nothing
#pragma unroll
                for (int k = 0; k < VEC; ++k) v[j][k] = 0.f;
                continue;
            }
            fx_load_lookup<VEC, SCALED>(a, p[j], d0, v[j]);
        }
#pragma unroll
        for (int j = 0; j < N; ++j)
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] += v[j][k];
    }
    return i;
}

// a run in chunks of INFL lookups, the remainder in halving chunks (INFL/2, ..., 1): at most
// log2(INFL) extra dependent rounds whatever the run length
template <int VEC, bool SCALED, int INFL>
__device__ __forceinline__ void fx_accum_run(const ReduceArgs& a, uint32_t beg, uint32_t end,
                                             uint32_t stride, int d0, float (&acc)[VEC]) {
    uint32_t i = fx_accum_chunks<VEC, SCALED, INFL>(a, beg, end, stride, d0, acc);
    if constexpr (INFL >= 16) i = fx_accum_chunks<VEC, SCALED, 8>(a, i, end, stride, d0, acc);
    if constexpr (INFL >= 8) i = fx_accum_chunks<VEC, SCALED, 4>(a, i, end, stride, d0, acc);
    if constexpr (INFL >= 4) i = fx_accum_chunks<VEC, SCALED, 2>(a, i, end, stride, d0, acc);
    if constexpr (INFL >= 2) fx_accum_chunks<VEC, SCALED, 1>(a, i, end, stride, d0, acc);
}

template <int VEC, bool SCALED, int INFL>
__global__ __launch_bounds__(256) void k_emb_grad_reduce_short(ReduceArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int rpb = 256 >> a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int nu = *a.n_unique;
    const int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2);
    if (u >= nu) return;
    const uint32_t beg = a.seg_start[u], end = a.seg_start[u + 1];
    if (end - beg > FX_LONG_RUN) {
        if (sub == 0) a.scratch[1 + atomicAdd(&a.scratch[0], 1)] = (int32_t)u;
        return;
    }
    if (d0 >= a.D) return;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    fx_accum_run<VEC, SCALED, INFL>(a, beg, end, 1u, d0, acc);
    fx_store<VEC>(a.G + u * a.D + d0, acc);
}

template <int VEC, bool SCALED, int INFL>
__global__ __launch_bounds__(256) void k_emb_grad_reduce_long(ReduceArgs a) {
    __shared__ float red[256 * VEC];
    const int lanes = 1 << a.lanes_log2;
    const int rpb = 256 >> a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int grp = threadIdx.x >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int n_long = a.scratch[0];
    for (int li = blockIdx.x; li < n_long; li += gridDim.x) {   // block-uniform
        const int64_t u = a.scratch[1 + li];
        const uint32_t beg = a.seg_start[u], end = a.seg_start[u + 1];
        float part[VEC];
#pragma unroll
        for (int k = 0; k < VEC; ++k) part[k] = 0.f;
        if (lane_on) fx_accum_run<VEC, SCALED, INFL>(a, beg + grp, end, (uint32_t)rpb, d0, part);
#pragma unroll
        for (int k = 0; k < VEC; ++k) red[k * 256 + threadIdx.x] = part[k];
        __syncthreads();
        for (int s = rpb >> 1; s > 0; s >>= 1) {
            if (grp < s) {
#pragma unroll
                for (int k = 0; k < VEC; ++k)
                    red[k * 256 + threadIdx.x] += red[k * 256 + threadIdx.x + (s << a.lanes_log2)];
            }
            __syncthreads();
        }
        if (grp == 0 && lane_on) {
            float out[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) out[k] = red[k * 256 + sub];
            fx_store<VEC>(a.G + u * a.D + d0, out);
        }
        __syncthreads();
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_rows_sqnorm(ReduceArgs a) {
    __shared__ float red4[4];
    const int lanes = 1 << a.lanes_log2;
    const int rpb = 256 >> a.lanes_log2;
    const int d0 = (threadIdx.x & (lanes - 1)) * VEC;
    const int nu = *a.n_unique;
    const int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2);
    float sq = 0.f;
    if (u < nu && d0 < a.D) {
        float g[VEC];
        fx_load<VEC>(a.G + u * a.D + d0, g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) sq = fmaf(g[k], g[k], sq);
    }
    const float tot = fx_block_sum_256(sq, red4);
    if (threadIdx.x == 0) {
        a.sq_partials[blockIdx.x] = tot;
        if (blockIdx.x == 0) a.scratch[0] = 0;   // ready for the next fx_emb_grad_reduce call
    }
}

extern "C" int64_t fx_emb_grad_reduce_partials(int64_t n_max, int32_t D) {
    if (n_max <= 0 || D < 1 || D > 256) return 1;
    return fx_ceil_div(n_max, 256 / fx_row_geom(D).lanes);
}

extern "C" int64_t fx_emb_grad_reduce_scratch_ints(int64_t n_max) {
    return (n_max <= 0 ? 0 : n_max / (FX_LONG_RUN + 1)) + 2;
}

extern "C" int fx_emb_grad_reduce_scaled(const float* dout, int64_t dout_ld,
                                         const int64_t* col_out_off, const int32_t* col_denom,
                                         const float* denom, int64_t denom_ld, int32_t C,
                                         int32_t D, const uint32_t* sorted_pos,
                                         const uint32_t* seg_start, const int32_t* n_unique,
                                         int64_t n_max, float* G, float* sq_partials,
                                         int32_t* scratch, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256 && C >= 1, "fx_emb_grad_reduce: bad C=%d / D=%d", C, D);
    hipStream_t s = fx_hip_stream(stream);
    FX_CHECK_ARG(sq_partials && scratch, "fx_emb_grad_reduce: null sq_partials / scratch");
    if (n_max <= 0) {
        hipLaunchKernelGGL(k_zero_words, dim3(1), dim3(64), 0, s, reinterpret_cast<int32_t*>(sq_partials), 1);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    FX_CHECK_ARG(dout && col_out_off && sorted_pos && seg_start && n_unique && G,
                 "fx_emb_grad_reduce: null pointer");
    const bool scaled = col_denom != nullptr;
    FX_CHECK_ARG(!scaled || denom, "fx_emb_grad_reduce_scaled: col_denom without denom");
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(dout_ld % g.vec == 0 || g.vec == 1,
                 "fx_emb_grad_reduce: dout_ld not a multiple of %d", g.vec);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    ReduceArgs a{dout, dout_ld, col_out_off, sorted_pos, seg_start, n_unique, G, sq_partials,
                 scratch, C, D, ll, col_denom, denom, denom_ld};
    // scratch[0] (the long-run counter) must be 0 on entry and is left 0 on return: the last launch
    // resets it (no memset node: as the ROOT node of a captured hipGraph segment a 4-byte
    // hipMemsetAsync was observed not to be ordered before the kernels that follow it)
    const int64_t blocks = fx_ceil_div(n_max, 256 / g.lanes);
    dim3 grid((unsigned)blocks), grid_long(512);
#define FX_REDUCE_LAUNCH_I(V, S, IS, IL)                                                     \
    hipLaunchKernelGGL((k_emb_grad_reduce_short<V, S, IS>), grid, dim3(256), 0, s, a);          \
    hipLaunchKernelGGL((k_emb_grad_reduce_long<V, S, IL>), grid_long, dim3(256), 0, s, a);      \
    hipLaunchKernelGGL(k_rows_sqnorm<V>, grid, dim3(256), 0, s, a);
    // lookups in flight per lane group: 4 for the short runs, 8 for the workgroup-per-row kernel
    // (scripts/reduce_bench.py: 31.5 -> 29.0 us for the three launches at D=16; deeper changes nothing)
#define FX_REDUCE_LAUNCH(V, S) FX_REDUCE_LAUNCH_I(V, S, 4, 8)
    if (scaled) {
        if (g.vec == 4) { FX_REDUCE_LAUNCH(4, true) }
        else if (g.vec == 2) { FX_REDUCE_LAUNCH(2, true) }
        else { FX_REDUCE_LAUNCH(1, true) }
    } else {
        if (g.vec == 4) { FX_REDUCE_LAUNCH(4, false) }
        else if (g.vec == 2) { FX_REDUCE_LAUNCH(2, false) }
        else { FX_REDUCE_LAUNCH(1, false) }
    }
#undef FX_REDUCE_LAUNCH_I
#undef FX_REDUCE_LAUNCH
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_emb_grad_reduce(const float* dout, int64_t dout_ld, const int64_t* col_out_off,
                                  int32_t C, int32_t D, const uint32_t* sorted_pos,
                                  const uint32_t* seg_start, const int32_t* n_unique,
                                  int64_t n_max, float* G, float* sq_partials, int32_t* scratch,
                                  fx_stream_t stream) {
    return fx_emb_grad_reduce_scaled(dout, dout_ld, col_out_off, nullptr, nullptr, 0, C, D,
                                     sorted_pos, seg_start, n_unique, n_max, G, sq_partials,
                                     scratch, stream);
}

// ---------------------------------------------------------------------------------------------
// row-sharded tables: routing of a rank's unique keys to their owners (fixed-capacity buckets so
// the all-to-all needs no host-side counts and the step stays free of host round trips)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int fx_lower_bound(const uint32_t* a, int n, uint32_t x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// e = o*cap + j over the padded send buffer: local row of the j-th unique key owned by rank o,
// or `pad_row` past the bucket's end
__global__ __launch_bounds__(256) void k_shard_send_idx(const uint32_t* uniq_key,
                                                        const int32_t* n_unique, int n_shards,
                                                        uint32_t rows_per_shard, int cap,
                                                        int32_t pad_row, int32_t* send_idx,
                                                        fx_scalars* scal) {
    const int nu = *n_unique;
    const int64_t total = (int64_t)n_shards * cap;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total;
         e += (int64_t)gridDim.x * 256) {
        const int o = (int)(e / cap), j = (int)(e - (int64_t)o * cap);
        const int s0 = fx_lower_bound(uniq_key, nu, (uint32_t)o * rows_per_shard);
        const int s1 = fx_lower_bound(uniq_key, nu, (uint32_t)(o + 1) * rows_per_shard);
        send_idx[e] = (j < s1 - s0) ? (int32_t)(uniq_key[s0 + j] - (uint32_t)o * rows_per_shard)
                                    : pad_row;
        if (j == 0 && s1 - s0 > cap) atomicOr(&scal->err_flag, FX_FLAG_A2A_OVERFLOW);
    }
}

// slot of unique key u inside the padded [n_shards*cap] buffers (pad slot on overflow)
__global__ __launch_bounds__(256) void k_shard_uniq_slot(const uint32_t* uniq_key,
                                                         const int32_t* n_unique, int n_shards,
                                                         uint32_t rows_per_shard, int cap,
                                                         int64_t n_max, int32_t* uniq_slot) {
    const int nu = *n_unique;
    for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < n_max;
         u += (int64_t)gridDim.x * 256) {
        int32_t slot = n_shards * cap;   // pad slot
        if (u < nu) {
            const uint32_t k = uniq_key[u];
            const int o = (int)(k / rows_per_shard);
            const int j = (int)u - fx_lower_bound(uniq_key, nu, (uint32_t)o * rows_per_shard);
            if (j < cap) slot = o * cap + j;
        }
        uniq_slot[u] = slot;
    }
}

// per lookup (b,c): where its row sits in the received-rows buffer
__global__ __launch_bounds__(256) void k_shard_lookup_slot(const uint32_t* sorted_pos,
                                                           const uint32_t* sorted_uid,
                                                           const int32_t* uniq_slot, int64_t n,
                                                           int32_t pad_slot, int32_t* lookup_slot) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t uid = sorted_uid[i];
        lookup_slot[sorted_pos[i]] = (uid == 0xFFFFFFFFu) ? pad_slot : uniq_slot[uid];
    }
}

// ---- plan for ROW-sorted unique keys (global rows g, owner = g % N, local row = g / N) ------------
// The column fast path of fx_dedup yields unique global rows in ascending order; owner-major order
// would need a device-wide sort (rocPRIM merge sort, ~60 us at 106 K keys).  Instead: per-owner
// ranks by counting — per-block owner histograms, a scan over the blocks, ranks inside a block from
// wave ballots.  Within an owner's bucket the keys stay in ascending row order (deterministic).
#define FX_PLAN_BLOCK 1024
#define FX_PLAN_MAX_SHARDS 64

__global__ __launch_bounds__(FX_PLAN_BLOCK) void k_shard_count(const uint32_t* uniq_key,
                                                               const int32_t* n_unique, int N,
                                                               int32_t* blk_cnt) {
    __shared__ int32_t cnt[FX_PLAN_MAX_SHARDS];
    if ((int)threadIdx.x < N) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int64_t u = (int64_t)blockIdx.x * FX_PLAN_BLOCK + threadIdx.x;
    if (u < *n_unique) atomicAdd(&cnt[uniq_key[u] % (uint32_t)N], 1);
    __syncthreads();
    if ((int)threadIdx.x < N) blk_cnt[(int64_t)blockIdx.x * N + threadIdx.x] = cnt[threadIdx.x];
}

// launch 2 of the plan.  Every block first turns the per-block owner histograms into ITS exclusive
// base per owner and the per-owner totals (<= a few hundred words, read by all blocks: the separate
// one-thread-per-owner scan launch of round 2 — 10 us — is gone), then
//   assigns      unique key u -> slot (owner, rank inside the owner's bucket); send_idx, uniq_slot, and
//                slot_uniq (slot -> u, the inverse map the gradient exchange fills its block by)
//   pads         bucket tails: send_idx = the owner's pad row, slot_uniq = -1
//   fills        lookup_slot with the pad slot (valid lookups are overwritten by launch 3)
//   flags        FX_FLAG_A2A_OVERFLOW when a bucket exceeds cap
// Round 2 did this in five launches (scan, assign, pad tails, fill, ...) of ~4.6 us each.
__global__ __launch_bounds__(FX_PLAN_BLOCK) void k_shard_route(const uint32_t* uniq_key,
                                                               const int32_t* n_unique, int N,
                                                               int cap, int64_t n_max,
                                                               const int32_t* blk_cnt, int nblk,
                                                               int32_t pad_row, int32_t* uniq_slot,
                                                               int32_t* send_idx, int32_t* slot_uniq,
                                                               int32_t* lookup_slot, fx_scalars* scal) {
    __shared__ int32_t wave_cnt[FX_PLAN_BLOCK / 64][FX_PLAN_MAX_SHARDS];
    __shared__ int32_t base_s[FX_PLAN_MAX_SHARDS], total_s[FX_PLAN_MAX_SHARDS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)threadIdx.x < N) {
        int32_t before = 0, all = 0;
        for (int b = 0; b < nblk; ++b) {
            const int32_t c = blk_cnt[(int64_t)b * N + threadIdx.x];
            if (b < (int)blockIdx.x) before += c;
            all += c;
        }
        base_s[threadIdx.x] = before;
        total_s[threadIdx.x] = all;
        if (blockIdx.x == 0 && all > cap) atomicOr(&scal->err_flag, FX_FLAG_A2A_OVERFLOW);
    }
    const int64_t u = (int64_t)blockIdx.x * FX_PLAN_BLOCK + threadIdx.x;
    const bool on = u < *n_unique;
    const uint32_t g = on ? uniq_key[u] : 0u;
    const int my_o = on ? (int)(g % (uint32_t)N) : -1;
    int my_rank = 0;
    for (int o = 0; o < N; ++o) {                           // block-uniform loop
        const unsigned long long m = __ballot(my_o == o);
        if (my_o == o) my_rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave][o] = __popcll(m);
    }
    __syncthreads();
    if (on) {
        int base = base_s[my_o];
        for (int w = 0; w < wave; ++w) base += wave_cnt[w][my_o];
        const int j = base + my_rank;
        int32_t slot = N * cap;                             // pad slot on overflow
        if (j < cap) {
            slot = my_o * cap + j;
            send_idx[slot] = (int32_t)(g / (uint32_t)N);
            if (slot_uniq) slot_uniq[slot] = (int32_t)u;
        }
        uniq_slot[u] = slot;
    } else if (u < n_max) {
        uniq_slot[u] = N * cap;
    }
    // bucket tails and the lookup-slot fill: grid-stride over the whole launch
    const int64_t total = (int64_t)N * cap;
    const int64_t tid = (int64_t)blockIdx.x * FX_PLAN_BLOCK + threadIdx.x;
    const int64_t nthr = (int64_t)gridDim.x * FX_PLAN_BLOCK;
    for (int64_t e = tid; e < total; e += nthr) {
        const int o = (int)(e / cap), j = (int)(e - (int64_t)o * cap);
        if (j >= total_s[o]) {
            send_idx[e] = pad_row;
            if (slot_uniq) slot_uniq[e] = -1;
        }
    }
    if (lookup_slot)
        for (int64_t i = tid; i < n_max; i += nthr) lookup_slot[i] = (int32_t)total;
}

// lookups whose id was padding / out of range have no position in the fast path's sorted arrays:
// every lookup slot starts at the pad slot (k_shard_route), valid positions are overwritten
__global__ __launch_bounds__(256) void k_shard_lookup_slot_g(const uint32_t* sorted_pos,
                                                             const uint32_t* sorted_uid,
                                                             const int32_t* uniq_slot, int64_t n,
                                                             int32_t* lookup_slot) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t p = sorted_pos[i], uid = sorted_uid[i];
        if (p != 0xFFFFFFFFu && uid != 0xFFFFFFFFu) lookup_slot[p] = uniq_slot[uid];
    }
}

__global__ __launch_bounds__(256) void k_fill_i32(int32_t* p, int64_t n, int32_t v) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        p[i] = v;
}

extern "C" int64_t fx_shard_plan_workspace_ints(int64_t n_lookups, int32_t n_shards) {
    if (n_lookups < 0 || n_shards < 1) return 0;
    return (fx_ceil_div(n_lookups > 0 ? n_lookups : 1, FX_PLAN_BLOCK) + 1) * (int64_t)n_shards;
}

extern "C" int fx_shard_plan(const uint32_t* uniq_key, const int32_t* n_unique,
                             const uint32_t* sorted_pos, const uint32_t* sorted_uid,
                             int64_t n_lookups, int32_t n_shards, int64_t total_rows, int32_t cap,
                             int32_t* send_idx, int32_t* uniq_slot, int32_t* lookup_slot,
                             fx_scalars* scal, int32_t global_keys, int32_t* workspace,
                             int32_t* slot_uniq, fx_stream_t stream) {
    FX_CHECK_ARG(n_shards >= 1 && cap >= 1 && n_lookups >= 0, "fx_shard_plan: bad sizes");
    FX_CHECK_ARG(uniq_key && n_unique && sorted_pos && sorted_uid && send_idx && uniq_slot &&
                     lookup_slot && scal,
                 "fx_shard_plan: null pointer");
    const int64_t rps = fx_ceil_div(total_rows, n_shards);
    hipStream_t s = fx_hip_stream(stream);
    const int64_t total = (int64_t)n_shards * cap;
    if (global_keys) {
        FX_CHECK_ARG(workspace, "fx_shard_plan: global_keys needs a workspace");
        FX_CHECK_ARG(n_shards <= FX_PLAN_MAX_SHARDS, "fx_shard_plan: n_shards=%d > %d", n_shards,
                     FX_PLAN_MAX_SHARDS);
        const int nblk = (int)fx_ceil_div(n_lookups > 0 ? n_lookups : 1, FX_PLAN_BLOCK);
        int32_t* blk_cnt = workspace;
        hipLaunchKernelGGL(k_shard_count, dim3((unsigned)nblk), dim3(FX_PLAN_BLOCK), 0, s, uniq_key,
                           n_unique, (int)n_shards, blk_cnt);
        hipLaunchKernelGGL(k_shard_route, dim3((unsigned)nblk), dim3(FX_PLAN_BLOCK), 0, s, uniq_key,
                           n_unique, (int)n_shards, (int)cap, n_lookups, blk_cnt, nblk, (int32_t)rps,
                           uniq_slot, send_idx, slot_uniq, lookup_slot, scal);
        if (n_lookups > 0) {
            int64_t b2 = fx_ceil_div(n_lookups, 256);
            if (b2 > 4096) b2 = 4096;
            hipLaunchKernelGGL(k_shard_lookup_slot_g, dim3((unsigned)b2), dim3(256), 0, s,
                               sorted_pos, sorted_uid, uniq_slot, n_lookups, lookup_slot);
        }
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    FX_CHECK_ARG(slot_uniq == nullptr, "fx_shard_plan: slot_uniq needs global_keys = 1");
    int64_t b1 = fx_ceil_div(total, 256);
    if (b1 > 4096) b1 = 4096;
    hipLaunchKernelGGL(k_shard_send_idx, dim3((unsigned)b1), dim3(256), 0, s, uniq_key, n_unique,
                       (int)n_shards, (uint32_t)rps, (int)cap, (int32_t)rps, send_idx, scal);
    if (n_lookups > 0) {
        int64_t b2 = fx_ceil_div(n_lookups, 256);
        if (b2 > 4096) b2 = 4096;
        hipLaunchKernelGGL(k_shard_uniq_slot, dim3((unsigned)b2), dim3(256), 0, s, uniq_key,
                           n_unique, (int)n_shards, (uint32_t)rps, (int)cap, n_lookups, uniq_slot);
        hipLaunchKernelGGL(k_shard_lookup_slot, dim3((unsigned)b2), dim3(256), 0, s, sorted_pos,
                           sorted_uid, uniq_slot, n_lookups, (int32_t)total, lookup_slot);
    }
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// dst[row_map[u] * dst_ld + 0..D) = src[u, :] for u < *n_rows (row gradients into their all-to-all
// slots; dst_ld > D when several table groups share one exchange block, each in its own columns)
template <int VEC>
__global__ __launch_bounds__(256) void k_scatter_rows(const float* src, const int32_t* row_map,
                                                      const int32_t* n_rows, int D, int lanes_log2,
                                                      float* dst, int64_t dst_ld) {
    const int lanes = 1 << lanes_log2;
    const int d0 = (threadIdx.x & (lanes - 1)) * VEC;
    const int64_t rpb = 256 >> lanes_log2;
    const int n = *n_rows;
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> lanes_log2); u < n;
         u += (int64_t)gridDim.x * rpb) {
        if (d0 >= D) continue;
        float v[VEC];
        fx_load<VEC>(src + u * D + d0, v);
        fx_store<VEC>(dst + (int64_t)row_map[u] * dst_ld + d0, v);
    }
}

extern "C" int fx_scatter_rows(const float* src, const int32_t* row_map, const int32_t* n_rows,
                               int64_t n_max, int32_t D, float* dst, int64_t dst_ld,
                               fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_scatter_rows: D=%d not in [1,256]", D);
    FX_CHECK_ARG(dst_ld >= D, "fx_scatter_rows: dst_ld < D");
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(src && row_map && n_rows && dst, "fx_scatter_rows: null pointer");
    FxRowGeom g = fx_row_geom(D);
    // vector stores need every destination row 16- / 8-byte aligned
    while (g.vec > 1 && (dst_ld % g.vec != 0 ||
                         (reinterpret_cast<uintptr_t>(dst) & (sizeof(float) * g.vec - 1)) != 0)) {
        g.vec >>= 1;
        g.lanes <<= 1;
    }
    FX_CHECK_ARG(g.lanes <= 256, "fx_scatter_rows: D=%d too wide for an unaligned destination", D);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    int64_t blocks = fx_ceil_div(n_max, 256 / g.lanes);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (g.vec == 4) hipLaunchKernelGGL(k_scatter_rows<4>, grid, dim3(256), 0, s, src, row_map, n_rows, (int)D, ll, dst, dst_ld);
    else if (g.vec == 2) hipLaunchKernelGGL(k_scatter_rows<2>, grid, dim3(256), 0, s, src, row_map, n_rows, (int)D, ll, dst, dst_ld);
    else hipLaunchKernelGGL(k_scatter_rows<1>, grid, dim3(256), 0, s, src, row_map, n_rows, (int)D, ll, dst, dst_ld);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// Gradient exchange of the row-sharded path, one launch each side (round 2: a zero fill + one
// scatter per table group before the all-to-all, three reduce launches per table group after it).
//   fx_fill_grad_block    requester: block[e, off_t .. off_t + D_t) = G_t[slot_uniq[e]] for every slot e
//                         of the [n_slots, ld] exchange block, zeros for empty slots / pad columns — the
//                         block is written once, densely (no zero fill, no scatter)
//   fx_owner_grad_reduce  owner: per unique owned row, the sum of the <= n_ranks received contributions
//                         (ascending rank order) of every table group + the squared-norm partials of all
//                         groups together (block b covers rows [b*256, (b+1)*256), fixed order)
// ---------------------------------------------------------------------------------------------
#define FX_GX_MAX_TABLES 4
struct GradBlockArgs {
    const float* G[FX_GX_MAX_TABLES];
    int32_t D[FX_GX_MAX_TABLES], off[FX_GX_MAX_TABLES];
    const int32_t* slot_uniq;
    float* block;
    int64_t ld, n_slots;
    int32_t n_tables;
};

__device__ __forceinline__ float fx_grad_block_elem(const GradBlockArgs& a, int32_t u, int c) {
    float v = 0.f;
#pragma unroll
    for (int t = 0; t < FX_GX_MAX_TABLES; ++t) {
        if (t < a.n_tables && a.G[t] != nullptr && c >= a.off[t] && c < a.off[t] + a.D[t])
            v = a.G[t][(int64_t)u * a.D[t] + (c - a.off[t])];
    }
    return v;
}

__global__ __launch_bounds__(256) void k_fill_grad_block(GradBlockArgs a) {
    const int64_t n = a.n_slots * a.ld;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i / a.ld;
        const int c = (int)(i - e * a.ld);
        const int32_t u = a.slot_uniq[e];
        a.block[i] = u >= 0 ? fx_grad_block_elem(a, u, c) : 0.f;
    }
}

// ld % 4 == 0 and a 16-byte aligned block: one thread per 4-float chunk of a slot's row — a chunk that
// lies inside one table group whose rows are float4-aligned is one 16-byte load, every chunk one
// 16-byte store (the [160 K, 20] block of c2 / c5: 11.9 -> ~5 us)
__global__ __launch_bounds__(256) void k_fill_grad_block_v4(GradBlockArgs a) {
    const int cpr = (int)(a.ld >> 2);                           // chunks per row
    const int64_t n = a.n_slots * cpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t e = i / cpr;
        const int c = (int)(i - e * cpr) * 4;
        const int32_t u = a.slot_uniq[e];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (u >= 0) {
            bool done = false;
#pragma unroll
            for (int t = 0; t < FX_GX_MAX_TABLES; ++t) {
                if (!done && t < a.n_tables && a.G[t] != nullptr && (a.D[t] & 3) == 0 &&
                    (a.off[t] & 3) == 0 && c >= a.off[t] && c + 4 <= a.off[t] + a.D[t]) {
                    v = *reinterpret_cast<const float4*>(a.G[t] + (int64_t)u * a.D[t] + (c - a.off[t]));
                    done = true;
                }
            }
            if (!done) {
                v.x = fx_grad_block_elem(a, u, c);
                v.y = fx_grad_block_elem(a, u, c + 1);
                v.z = fx_grad_block_elem(a, u, c + 2);
                v.w = fx_grad_block_elem(a, u, c + 3);
            }
        }
        *reinterpret_cast<float4*>(a.block + e * a.ld + c) = v;
    }
}

extern "C" int fx_fill_grad_block(const float* const* G_host, const int32_t* D_host,
                                  const int32_t* off_host, int32_t n_tables,
                                  const int32_t* slot_uniq, int64_t n_slots, float* block, int64_t ld,
                                  fx_stream_t stream) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_GX_MAX_TABLES, "fx_fill_grad_block: n_tables=%d not in [1,%d]",
                 n_tables, FX_GX_MAX_TABLES);
    FX_CHECK_ARG(n_slots >= 0 && ld >= 1, "fx_fill_grad_block: bad sizes");
    if (n_slots == 0) return FX_OK;
    FX_CHECK_ARG(G_host && D_host && off_host && slot_uniq && block, "fx_fill_grad_block: null pointer");
    GradBlockArgs a;
    memset(&a, 0, sizeof(a));
    for (int t = 0; t < n_tables; ++t) {
        FX_CHECK_ARG(D_host[t] >= 1 && off_host[t] >= 0 && off_host[t] + D_host[t] <= ld,
                     "fx_fill_grad_block: table %d does not fit the block", t);
        a.G[t] = G_host[t];                 // NULL: this group has no gradient this step (zeros)
        a.D[t] = D_host[t];
        a.off[t] = off_host[t];
    }
    a.slot_uniq = slot_uniq; a.block = block; a.ld = ld; a.n_slots = n_slots; a.n_tables = n_tables;
    bool v4 = ld % 4 == 0 && (reinterpret_cast<uintptr_t>(block) & 15) == 0;
    for (int t = 0; t < n_tables; ++t)
        if (a.G[t] && (a.D[t] & 3) == 0 && (reinterpret_cast<uintptr_t>(a.G[t]) & 15) != 0) v4 = false;
    int64_t blocks = fx_ceil_div(v4 ? n_slots * (ld / 4) : n_slots * ld, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (v4) hipLaunchKernelGGL(k_fill_grad_block_v4, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    else hipLaunchKernelGGL(k_fill_grad_block, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

#define FX_OGR_ROWS 32       // unique rows per workgroup (= per squared-norm partial)
#define FX_OGR_PASSES 4      // rows a thread serves, their loads in flight together
struct OwnerReduceArgs {
    float* G[FX_GX_MAX_TABLES];
    int32_t D[FX_GX_MAX_TABLES], off[FX_GX_MAX_TABLES];
    const float* grecv;
    int64_t ld;
    const uint32_t* sorted_pos;
    const uint32_t* seg_start;
    const int32_t* n_unique;
    float* sq_partials;
    int32_t n_tables, width;     // width = floats of a row that carry gradients
};

// 32 lanes per unique row (lane c = column c of the row's gradient columns, all groups side by side;
// rows wider than 32 floats take several column passes), 8 rows per pass of the workgroup, 4 passes
// whose loads are issued together.  A row's run has at most n_ranks entries.
__global__ __launch_bounds__(256) void k_owner_grad_reduce(OwnerReduceArgs a) {
    __shared__ float red[4];
    const int c_lane = threadIdx.x & 31, r_in = threadIdx.x >> 5;
    const int nu = *a.n_unique;
    float sq = 0.f;
    uint32_t beg[FX_OGR_PASSES], len[FX_OGR_PASSES];
    int64_t u[FX_OGR_PASSES];
    uint32_t maxlen = 0;
#pragma unroll
    for (int p = 0; p < FX_OGR_PASSES; ++p) {
        u[p] = (int64_t)blockIdx.x * FX_OGR_ROWS + p * 8 + r_in;
        beg[p] = len[p] = 0;
        if (u[p] < nu) {
            beg[p] = a.seg_start[u[p]];
            len[p] = a.seg_start[u[p] + 1] - beg[p];
        }
        maxlen = len[p] > maxlen ? len[p] : maxlen;
    }
    for (int c0 = 0; c0 < a.width; c0 += 32) {
        const int c = c0 + c_lane;
        int t = -1;
#pragma unroll
        for (int k = 0; k < FX_GX_MAX_TABLES; ++k)
            if (k < a.n_tables && a.G[k] != nullptr && c >= a.off[k] && c < a.off[k] + a.D[k]) t = k;
        float acc[FX_OGR_PASSES];
#pragma unroll
        for (int p = 0; p < FX_OGR_PASSES; ++p) acc[p] = 0.f;
        if (t >= 0) {
            for (uint32_t j = 0; j < maxlen; ++j) {            // ascending rank order: deterministic
                float v[FX_OGR_PASSES];
#pragma unroll
                for (int p = 0; p < FX_OGR_PASSES; ++p) {
                    v[p] = 0.f;
                    if (j < len[p]) v[p] = a.grecv[(int64_t)a.sorted_pos[beg[p] + j] * a.ld + c];
                }
#pragma unroll
                for (int p = 0; p < FX_OGR_PASSES; ++p) acc[p] += v[p];
            }
#pragma unroll
            for (int p = 0; p < FX_OGR_PASSES; ++p)
                if (u[p] < nu) a.G[t][u[p] * a.D[t] + (c - a.off[t])] = acc[p];
        }
#pragma unroll
        for (int p = 0; p < FX_OGR_PASSES; ++p) sq = fmaf(acc[p], acc[p], sq);
    }
    const float tot = fx_block_sum_256(sq, red);
    if (threadIdx.x == 0) a.sq_partials[blockIdx.x] = tot;
}

extern "C" int64_t fx_owner_grad_reduce_partials(int64_t n_max) {
    return n_max <= 0 ? 1 : fx_ceil_div(n_max, FX_OGR_ROWS);
}

extern "C" int fx_owner_grad_reduce(const float* grecv, int64_t ld, const uint32_t* sorted_pos,
                                    const uint32_t* seg_start, const int32_t* n_unique, int64_t n_max,
                                    float* const* G_host, const int32_t* D_host,
                                    const int32_t* off_host, int32_t n_tables, float* sq_partials,
                                    fx_stream_t stream) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_GX_MAX_TABLES,
                 "fx_owner_grad_reduce: n_tables=%d not in [1,%d]", n_tables, FX_GX_MAX_TABLES);
    FX_CHECK_ARG(sq_partials, "fx_owner_grad_reduce: null sq_partials");
    if (n_max <= 0) {
        hipLaunchKernelGGL(k_fill_i32, dim3(1), dim3(256), 0, fx_hip_stream(stream),
                           reinterpret_cast<int32_t*>(sq_partials), (int64_t)1, 0);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    FX_CHECK_ARG(grecv && sorted_pos && seg_start && n_unique && G_host && D_host && off_host,
                 "fx_owner_grad_reduce: null pointer");
    OwnerReduceArgs a;
    memset(&a, 0, sizeof(a));
    int width = 0;
    for (int t = 0; t < n_tables; ++t) {
        FX_CHECK_ARG(D_host[t] >= 1 && off_host[t] >= 0 && off_host[t] + D_host[t] <= ld,
                     "fx_owner_grad_reduce: table %d does not fit the block", t);
        a.G[t] = G_host[t];                 // NULL: no gradient for this group this step
        a.D[t] = D_host[t];
        a.off[t] = off_host[t];
        if (off_host[t] + D_host[t] > width) width = off_host[t] + D_host[t];
    }
    a.grecv = grecv; a.ld = ld; a.sorted_pos = sorted_pos; a.seg_start = seg_start;
    a.n_unique = n_unique; a.sq_partials = sq_partials; a.n_tables = n_tables; a.width = width;
    const int64_t blocks = fx_ceil_div(n_max, FX_OGR_ROWS);
    hipLaunchKernelGGL(k_owner_grad_reduce, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// The block of rows an all-to-all delivered, [n_rows, src_ld] with one column range per table group,
// into one contiguous [n_rows + zero_tail_rows, width_p] buffer per group (the gather kernels read
// rows of exactly D floats); the tail rows — the pad slot padding lookups point at — are zeroed.
#define FX_SPLIT_MAX_PARTS 4
struct SplitArgs {
    const float* src;
    int64_t src_ld, n_rows;
    int32_t n_parts, tail, total_w;
    float* dst[FX_SPLIT_MAX_PARTS];
    int32_t off[FX_SPLIT_MAX_PARTS], w[FX_SPLIT_MAX_PARTS];
};

__global__ __launch_bounds__(256) void k_split_rows(SplitArgs a) {
    const int64_t n = a.n_rows * a.total_w, nt = (int64_t)a.tail * a.total_w;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n + nt;
         i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / a.total_w;
        int c = (int)(i - r * a.total_w);
        int p = 0;
        while (c >= a.w[p]) c -= a.w[p++];                 // part p, column c of it
        a.dst[p][r * a.w[p] + c] = r < a.n_rows ? a.src[r * a.src_ld + a.off[p] + c] : 0.f;
    }
}

extern "C" int fx_split_rows(const float* src, int64_t src_ld, int64_t n_rows, int32_t n_parts,
                             float* const* dst_host, const int32_t* off_host,
                             const int32_t* width_host, int32_t zero_tail_rows, fx_stream_t stream) {
    FX_CHECK_ARG(n_parts >= 1 && n_parts <= FX_SPLIT_MAX_PARTS && n_rows >= 0 && zero_tail_rows >= 0,
                 "fx_split_rows: bad sizes (1..%d parts)", FX_SPLIT_MAX_PARTS);
    FX_CHECK_ARG(src && dst_host && off_host && width_host, "fx_split_rows: null pointer");
    SplitArgs a;
    memset(&a, 0, sizeof(a));
    a.src = src; a.src_ld = src_ld; a.n_rows = n_rows; a.n_parts = n_parts; a.tail = zero_tail_rows;
    for (int p = 0; p < n_parts; ++p) {
        FX_CHECK_ARG(dst_host[p] && width_host[p] >= 1 && off_host[p] >= 0 &&
                     off_host[p] + width_host[p] <= src_ld, "fx_split_rows: bad part %d", p);
        a.dst[p] = dst_host[p]; a.off[p] = off_host[p]; a.w[p] = width_host[p];
        a.total_w += width_host[p];
    }
    const int64_t total = (n_rows + zero_tail_rows) * a.total_w;
    if (total == 0) return FX_OK;
    int64_t blocks = fx_ceil_div(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_split_rows, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}


// ---------------------------------------------------------------------------------------------
// optimizer scalars
// ---------------------------------------------------------------------------------------------
__global__ void k_opt_begin_step(fx_scalars* sc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) fx_begin_step_dev(sc);
}

extern "C" int fx_opt_begin_step(fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(scal, "fx_opt_begin_step: null scal");
    hipLaunchKernelGGL(k_opt_begin_step, dim3(1), dim3(64), 0, fx_hip_stream(stream), scal);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

#define FX_CLIP_MAX_PARTS 16
struct ClipArgs {
    const float* part[FX_CLIP_MAX_PARTS];
    int64_t count[FX_CLIP_MAX_PARTS];
    int32_t n_parts;
    fx_scalars* scal;
};

__global__ __launch_bounds__(1024) void k_clip_coef(ClipArgs a) {
    __shared__ double red[1024];
    // fixed assignment thread <- elements t, t+1024, ... of every array, in array order: deterministic
    double acc = 0.0;
    for (int p = 0; p < a.n_parts; ++p) {
        const float* x = a.part[p];
        for (int64_t i = threadIdx.x; i < a.count[p]; i += 1024) acc += (double)x[i];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0]);
        float coef = 1.f;
        if (a.scal->max_norm > 0.f) {
            coef = a.scal->max_norm / (total + 1e-6f);
            if (coef > 1.f) coef = 1.f;
        }
        a.scal->total_norm = total;
        a.scal->clip_coef = coef;
    }
}

// out[0] = sum of the given partial arrays (fixed order): the rank-local table part of the
// global gradient norm, all-reduced by the host before fx_clip_coef
__global__ __launch_bounds__(1024) void k_sum_parts(ClipArgs a, float* out) {
    __shared__ double red[1024];
    double acc = 0.0;
    for (int p = 0; p < a.n_parts; ++p) {
        const float* x = a.part[p];
        for (int64_t i = threadIdx.x; i < a.count[p]; i += 1024) acc += (double)x[i];
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

extern "C" int fx_sum_parts(const float* const* parts_host, const int64_t* counts_host,
                            int32_t n_parts, float* out, fx_stream_t stream) {
    FX_CHECK_ARG(n_parts >= 0 && n_parts <= FX_CLIP_MAX_PARTS, "fx_sum_parts: n_parts=%d > %d",
                 n_parts, FX_CLIP_MAX_PARTS);
    FX_CHECK_ARG(out, "fx_sum_parts: null out");
    ClipArgs a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < n_parts; ++p) {
        FX_CHECK_ARG(parts_host[p] || counts_host[p] == 0, "fx_sum_parts: null part %d", p);
        a.part[p] = parts_host[p];
        a.count[p] = counts_host[p];
    }
    a.n_parts = n_parts;
    hipLaunchKernelGGL(k_sum_parts, dim3(1), dim3(1024), 0, fx_hip_stream(stream), a, out);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_clip_coef(const float* const* parts_host, const int64_t* counts_host,
                            int32_t n_parts, fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(n_parts >= 0 && n_parts <= FX_CLIP_MAX_PARTS, "fx_clip_coef: n_parts=%d > %d",
                 n_parts, FX_CLIP_MAX_PARTS);
    FX_CHECK_ARG(scal, "fx_clip_coef: null scal");
    ClipArgs a;
    memset(&a, 0, sizeof(a));
    for (int p = 0; p < n_parts; ++p) {
        FX_CHECK_ARG(parts_host[p] || counts_host[p] == 0, "fx_clip_coef: null part %d", p);
        a.part[p] = parts_host[p];
        a.count[p] = counts_host[p];
    }
    a.n_parts = n_parts;
    a.scal = scal;
    hipLaunchKernelGGL(k_clip_coef, dim3(1), dim3(1024), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// sparse-row optimizers
// ---------------------------------------------------------------------------------------------
struct RowOptArgs {
    float* table;
    float* m;
    float* v;
    int32_t* last_step;
    const uint32_t* uniq_row;
    const int32_t* n_unique;
    const float* G;
    const fx_scalars* scal;
    int64_t total_rows;
    int32_t D, lanes_log2, upto_offset;
};

// torch.optim.Adam._single_tensor_adam, one element
__device__ __forceinline__ void fx_adam_elem(float& p, float& m, float& v, float g, float w1,
                                             float beta2, float w2, float bc2_sqrt, float eps,
                                             float step_size) {
    m = m + w1 * (g - m);                  // exp_avg.lerp_(grad, 1 - beta1)
    v = fmaf(w2 * g, g, v * beta2);        // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    p = p - step_size * (m / denom);       // param.addcdiv_(exp_avg, denom, value=-step_size)
}

// d/dp of (l1 * |p| + l2/2 * p^2): the embedding regularizer of rank_model.py:106-112
__device__ __forceinline__ float fx_reg_grad(float p, float l1, float l2) {
    float r = l2 * p;
    if (l1 != 0.f) r += p > 0.f ? l1 : (p < 0.f ? -l1 : 0.f);
    return r;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_sparse_adam(RowOptArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int nu = *a.n_unique;
    const int64_t rpb = 256 >> a.lanes_log2;
    const fx_scalars sc = *a.scal;
    const float w1 = fx_one_minus(sc.beta1), w2 = fx_one_minus(sc.beta2);   // torch's float(1 - beta)
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2); u < nu;
         u += (int64_t)gridDim.x * rpb) {
        const int64_t row = a.uniq_row[u];
        if (d0 < a.D) {
            float p[VEC], m[VEC], v[VEC], g[VEC];
            const int64_t o = row * a.D + d0;
            fx_load<VEC>(a.table + o, p);
            fx_load<VEC>(a.m + o, m);
            fx_load<VEC>(a.v + o, v);
            fx_load<VEC>(a.G + u * a.D + d0, g);
            if (sc.reg_l1 != 0.f || sc.reg_l2 != 0.f) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) g[k] += fx_reg_grad(p[k], sc.reg_l1, sc.reg_l2);
            }
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                fx_adam_elem(p[k], m[k], v[k], g[k] * sc.clip_coef, w1, sc.beta2, w2, sc.bc2_sqrt,
                             sc.eps, sc.step_size);
            fx_store<VEC>(a.table + o, p);
            fx_store<VEC>(a.m + o, m);
            fx_store<VEC>(a.v + o, v);
        }
        if (sub == 0 && a.last_step) a.last_step[row] = sc.step;
    }
}

// Replay of the zero-gradient Adam steps a dense optimizer would have applied to a row that no
// batch touched since last_step[row].  With g = 0 the per-step update is
//     m *= beta1 ; v *= beta2 ; p -= lr/(1-beta1^j) * m / (sqrt(v)/sqrt(1-beta2^j) + eps)
// whose magnitude decays like (beta1/sqrt(beta2))^j ~ 0.9^j, so after FX_REPLAY_MAX steps the
// remaining terms are below fp32 resolution of the accumulated update; the tail only decays m, v
// (fx_adam_replay in fx_common.h).

template <int VEC>
__global__ __launch_bounds__(256) void k_adam_catchup(RowOptArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int64_t rpb = 256 >> a.lanes_log2;
    const fx_scalars sc = *a.scal;
    const int upto = sc.step + a.upto_offset;
    const int64_t n = a.uniq_row ? (int64_t)(*a.n_unique) : a.total_rows;
    const FxLogs lg = fx_logs_of(sc);
    const FxSeries ser = fx_series_of(a.scal, sc);
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2); u < n;
         u += (int64_t)gridDim.x * rpb) {
        const int64_t row = a.uniq_row ? (int64_t)a.uniq_row[u] : u;
        const int last = a.last_step[row];
        const int k_steps = upto - last;
        if (k_steps <= 0) continue;
        if (d0 < a.D) {
            float p[VEC], m[VEC], v[VEC];
            const int64_t o = row * a.D + d0;
            fx_load<VEC>(a.m + o, m);
            fx_load<VEC>(a.v + o, v);
            bool any = false;
#pragma unroll
            for (int k = 0; k < VEC; ++k) any = any || (m[k] != 0.f) || (v[k] != 0.f);
            if (any) {
                fx_load<VEC>(a.table + o, p);
                fx_adam_replay<VEC>(p, m, v, last, k_steps, sc, lg, ser);
                fx_store<VEC>(a.table + o, p);
                fx_store<VEC>(a.m + o, m);
                fx_store<VEC>(a.v + o, v);
            }
        }
        if (sub == 0) a.last_step[row] = upto;
    }
}

template <int VEC>
__global__ __launch_bounds__(256) void k_sparse_sgd(RowOptArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int nu = *a.n_unique;
    const int64_t rpb = 256 >> a.lanes_log2;
    const float scale = a.scal->lr * a.scal->clip_coef;
    const float l1 = a.scal->reg_l1, l2 = a.scal->reg_l2;
    const int step = a.scal->step;
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2); u < nu;
         u += (int64_t)gridDim.x * rpb) {
        if (d0 >= a.D) continue;
        const int64_t row = a.uniq_row[u];
        float p[VEC], g[VEC];
        fx_load<VEC>(a.table + row * a.D + d0, p);
        fx_load<VEC>(a.G + u * a.D + d0, g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) p[k] = p[k] - scale * (g[k] + fx_reg_grad(p[k], l1, l2));
        fx_store<VEC>(a.table + row * a.D + d0, p);
        if (sub == 0 && a.last_step) a.last_step[row] = step;
    }
}

// ---------------------------------------------------------------------------------------------
// embedding regularizer (rank_model.py:95-112): a dense term over EVERY table row.  Three kernels:
//   k_reg_stats   sum p^2, sum |p|, sum r^2 (r = l1 sign(p) + l2 p) over the whole packed table
//   k_reg_cross   sum 2 G.r over the rows the batch touched: with it
//                 |G + r|^2 summed over all rows = sum r^2 + sum G^2 + sum 2 G.r
//   k_reg_dense   the optimizer step with g = r for the rows the batch did NOT touch
//                 (touched rows: k_sparse_adam / k_sparse_sgd add r themselves and mark last_step)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reg_stats(const float* __restrict__ x, int64_t n,
                                                   const fx_scalars* scal, float* partials) {
    __shared__ float red4[4];
    const float l1 = scal->reg_l1, l2 = scal->reg_l2;
    float s2 = 0.f, s1 = 0.f, sr = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? (n >> 2) : 0;
    const float4* x4 = reinterpret_cast<const float4*>(x);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const float4 q = x4[i];
        const float e[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float r = fx_reg_grad(e[k], l1, l2);
            s2 = fmaf(e[k], e[k], s2);
            s1 += fabsf(e[k]);
            sr = fmaf(r, r, sr);
        }
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const float e = x[i], r = fx_reg_grad(e, l1, l2);
        s2 = fmaf(e, e, s2);
        s1 += fabsf(e);
        sr = fmaf(r, r, sr);
    }
    const float t2 = fx_block_sum_256(s2, red4);
    __syncthreads();
    const float t1 = fx_block_sum_256(s1, red4);
    __syncthreads();
    const float tr = fx_block_sum_256(sr, red4);
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = t2;
        partials[FX_REG_BLOCKS + blockIdx.x] = t1;
        partials[2 * FX_REG_BLOCKS + blockIdx.x] = tr;
    }
}

extern "C" int fx_reg_stats(const float* x, int64_t n, const fx_scalars* scal, float* partials,
                            fx_stream_t stream) {
    FX_CHECK_ARG(n >= 0, "fx_reg_stats: n=%lld", (long long)n);
    FX_CHECK_ARG((x || n == 0) && scal && partials, "fx_reg_stats: null pointer");
    hipLaunchKernelGGL(k_reg_stats, dim3(FX_REG_BLOCKS), dim3(256), 0, fx_hip_stream(stream), x, n,
                       scal, partials);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_reg_cross(RowOptArgs a, float* partials) {
    __shared__ float red4[4];
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int nu = *a.n_unique;
    const int64_t rpb = 256 >> a.lanes_log2;
    const float l1 = a.scal->reg_l1, l2 = a.scal->reg_l2;
    float acc = 0.f;
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2); u < nu;
         u += (int64_t)gridDim.x * rpb) {
        if (d0 >= a.D) continue;
        const int64_t row = a.uniq_row[u];
        float p[VEC], g[VEC];
        fx_load<VEC>(a.table + row * a.D + d0, p);
        fx_load<VEC>(a.G + u * a.D + d0, g);
#pragma unroll
        for (int k = 0; k < VEC; ++k) acc = fmaf(2.f * g[k], fx_reg_grad(p[k], l1, l2), acc);
    }
    const float tot = fx_block_sum_256(acc, red4);
    if (threadIdx.x == 0) partials[blockIdx.x] = tot;
}

extern "C" int fx_reg_cross(const float* table, int32_t D, const uint32_t* uniq_row,
                            const int32_t* n_unique, int64_t n_max, const float* G,
                            const fx_scalars* scal, float* partials, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_reg_cross: D=%d not in [1,256]", D);
    FX_CHECK_ARG(table && uniq_row && n_unique && G && scal && partials,
                 "fx_reg_cross: null pointer");
    RowOptArgs a{const_cast<float*>(table), nullptr, nullptr, nullptr, uniq_row, n_unique, G, scal,
                 0, D, 0, 0};
    const FxRowGeom g = fx_row_geom(D);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    a.lanes_log2 = ll;
    hipStream_t s = fx_hip_stream(stream);
    dim3 grid(FX_REG_CROSS_BLOCKS);       // fixed: every block writes its (possibly zero) partial
    if (g.vec == 4) hipLaunchKernelGGL(k_reg_cross<4>, grid, dim3(256), 0, s, a, partials);
    else if (g.vec == 2) hipLaunchKernelGGL(k_reg_cross<2>, grid, dim3(256), 0, s, a, partials);
    else hipLaunchKernelGGL(k_reg_cross<1>, grid, dim3(256), 0, s, a, partials);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

template <int VEC, bool ADAM>
__global__ __launch_bounds__(256) void k_reg_dense(RowOptArgs a) {
    const int lanes = 1 << a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const int64_t rpb = 256 >> a.lanes_log2;
    const fx_scalars sc = *a.scal;
    const float w1 = fx_one_minus(sc.beta1), w2 = fx_one_minus(sc.beta2);   // torch's float(1 - beta)
    const float scale = sc.lr * sc.clip_coef;
    for (int64_t row = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.lanes_log2); row < a.total_rows;
         row += (int64_t)gridDim.x * rpb) {
        if (d0 >= a.D) continue;
        if (a.last_step[row] == sc.step) continue;       // updated by the sparse kernel this step
        const int64_t o = row * a.D + d0;
        float p[VEC];
        fx_load<VEC>(a.table + o, p);
        if constexpr (ADAM) {
            float m[VEC], v[VEC];
            fx_load<VEC>(a.m + o, m);
            fx_load<VEC>(a.v + o, v);
#pragma unroll
            for (int k = 0; k < VEC; ++k)
                fx_adam_elem(p[k], m[k], v[k], fx_reg_grad(p[k], sc.reg_l1, sc.reg_l2) * sc.clip_coef,
                             w1, sc.beta2, w2, sc.bc2_sqrt, sc.eps, sc.step_size);
            fx_store<VEC>(a.m + o, m);
            fx_store<VEC>(a.v + o, v);
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) p[k] = p[k] - scale * fx_reg_grad(p[k], sc.reg_l1, sc.reg_l2);
        }
        fx_store<VEC>(a.table + o, p);
    }
}

extern "C" int fx_reg_dense_update(float* table, float* m, float* v, const int32_t* last_step,
                                   int64_t total_rows, int32_t D, int32_t adam,
                                   const fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_reg_dense_update: D=%d not in [1,256]", D);
    if (total_rows <= 0) return FX_OK;
    FX_CHECK_ARG(table && last_step && scal && (!adam || (m && v)),
                 "fx_reg_dense_update: null pointer");
    RowOptArgs a{table, m, v, const_cast<int32_t*>(last_step), nullptr, nullptr, nullptr, scal,
                 total_rows, D, 0, 0};
    const FxRowGeom g = fx_row_geom(D);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    a.lanes_log2 = ll;
    int64_t blocks = fx_ceil_div(total_rows, 256 / g.lanes);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
#define FX_REG_DENSE(V)                                                                      \
    do {                                                                                     \
        if (adam) hipLaunchKernelGGL((k_reg_dense<V, true>), grid, dim3(256), 0, s, a);      \
        else hipLaunchKernelGGL((k_reg_dense<V, false>), grid, dim3(256), 0, s, a);          \
    } while (0)
    if (g.vec == 4) FX_REG_DENSE(4);
    else if (g.vec == 2) FX_REG_DENSE(2);
    else FX_REG_DENSE(1);
#undef FX_REG_DENSE
    FX_CHECK_LAUNCH();
    return FX_OK;
}

#define FX_LAUNCH_ROWOPT(KERNEL, n_rows_max)                                             \
    do {                                                                                 \
        const FxRowGeom g = fx_row_geom(D);                                              \
        int ll = 0;                                                                      \
        while ((1 << ll) < g.lanes) ++ll;                                                \
        a.lanes_log2 = ll;                                                               \
        int64_t blocks = fx_ceil_div((n_rows_max), 256 / g.lanes);                       \
        if (blocks > 256 * 64) blocks = 256 * 64;                                        \
        if (blocks < 1) blocks = 1;                                                      \
        dim3 grid((unsigned)blocks);                                                     \
        hipStream_t s = fx_hip_stream(stream);                                           \
        if (g.vec == 4) hipLaunchKernelGGL(KERNEL<4>, grid, dim3(256), 0, s, a);         \
        else if (g.vec == 2) hipLaunchKernelGGL(KERNEL<2>, grid, dim3(256), 0, s, a);    \
        else hipLaunchKernelGGL(KERNEL<1>, grid, dim3(256), 0, s, a);                    \
        FX_CHECK_LAUNCH();                                                               \
    } while (0)

extern "C" int fx_sparse_adam(float* table, float* m, float* v, int32_t* last_step, int32_t D,
                              const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                              const float* G, const fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_sparse_adam: D=%d not in [1,256]", D);
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(table && m && v && uniq_row && n_unique && G && scal,
                 "fx_sparse_adam: null pointer");
    RowOptArgs a{table, m, v, last_step, uniq_row, n_unique, G, scal, 0, D, 0, 0};
    FX_LAUNCH_ROWOPT(k_sparse_adam, n_max);
    return FX_OK;
}

extern "C" int fx_adam_catchup(float* table, float* m, float* v, int32_t* last_step, int32_t D,
                               const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                               int64_t total_rows, int32_t upto_offset, const fx_scalars* scal,
                               fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_adam_catchup: D=%d not in [1,256]", D);
    FX_CHECK_ARG(table && m && v && last_step && scal, "fx_adam_catchup: null pointer");
    FX_CHECK_ARG(uniq_row == nullptr || n_unique != nullptr,
                 "fx_adam_catchup: uniq_row given without n_unique");
    const int64_t rows = uniq_row ? n_max : total_rows;
    if (rows <= 0) return FX_OK;
    RowOptArgs a{table, m, v, last_step, uniq_row, n_unique, nullptr, scal, total_rows, D, 0,
                 upto_offset};
    FX_LAUNCH_ROWOPT(k_adam_catchup, rows);
    return FX_OK;
}

extern "C" int fx_sparse_sgd(float* table, int32_t* last_step, int32_t D, const uint32_t* uniq_row,
                             const int32_t* n_unique, int64_t n_max, const float* G,
                             const fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_sparse_sgd: D=%d not in [1,256]", D);
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(table && uniq_row && n_unique && G && scal, "fx_sparse_sgd: null pointer");
    RowOptArgs a{table, nullptr, nullptr, last_step, uniq_row, n_unique, G, scal, 0, D, 0, 0};
    FX_LAUNCH_ROWOPT(k_sparse_sgd, n_max);
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-tensor dense kernels (MLP / CrossNet / bias / numeric-weight parameters)
// ---------------------------------------------------------------------------------------------
#define FX_MT_MAX 64
struct MtArgs {
    float* p[FX_MT_MAX];
    const float* g[FX_MT_MAX];
    float* m[FX_MT_MAX];
    float* v[FX_MT_MAX];
    int64_t size[FX_MT_MAX];
    const fx_scalars* scal;
    float* sq_partials;
};

__global__ __launch_bounds__(256) void k_mt_sqnorm(MtArgs a) {
    __shared__ float red4[4];
    const int t = blockIdx.y;
    const float* g = a.g[t];
    const int64_t n = a.size[t];
    float acc = 0.f;
    const int64_t stride = (int64_t)gridDim.x * 256;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            const float4 x = g4[i];
            acc = fmaf(x.x, x.x, acc);
            acc = fmaf(x.y, x.y, acc);
            acc = fmaf(x.z, x.z, acc);
            acc = fmaf(x.w, x.w, acc);
        }
        for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
            acc = fmaf(g[i], g[i], acc);
    } else {
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
            acc = fmaf(g[i], g[i], acc);
    }
    const float tot = fx_block_sum_256(acc, red4);
    if (threadIdx.x == 0) a.sq_partials[(int64_t)t * gridDim.x + blockIdx.x] = tot;
}

__global__ __launch_bounds__(256) void k_mt_adam(MtArgs a) {
    const int t = blockIdx.y;
    float* p = a.p[t];
    const float* g = a.g[t];
    float* m = a.m[t];
    float* v = a.v[t];
    const int64_t n = a.size[t];
    const fx_scalars sc = *a.scal;
    const float w1 = fx_one_minus(sc.beta1), w2 = fx_one_minus(sc.beta2);   // torch's float(1 - beta)
    const int64_t stride = (int64_t)gridDim.x * 256;
    const uintptr_t al = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) |
                         reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v);
    int64_t done = 0;
    if ((al & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
            float pp[4], gg[4], mm[4], vv[4];
            fx_load<4>(p + 4 * i, pp);
            fx_load<4>(g + 4 * i, gg);
            fx_load<4>(m + 4 * i, mm);
            fx_load<4>(v + 4 * i, vv);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                fx_adam_elem(pp[k], mm[k], vv[k], gg[k] * sc.clip_coef, w1, sc.beta2, w2,
                             sc.bc2_sqrt, sc.eps, sc.step_size);
            fx_store<4>(p + 4 * i, pp);
            fx_store<4>(m + 4 * i, mm);
            fx_store<4>(v + 4 * i, vv);
        }
        done = n4 << 2;
    }
    for (int64_t i = done + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float pp = p[i], mm = m[i], vv = v[i];
        fx_adam_elem(pp, mm, vv, g[i] * sc.clip_coef, w1, sc.beta2, w2, sc.bc2_sqrt, sc.eps,
                     sc.step_size);
        p[i] = pp;
        m[i] = mm;
        v[i] = vv;
    }
}

__global__ __launch_bounds__(256) void k_mt_sgd(MtArgs a) {
    const int t = blockIdx.y;
    float* p = a.p[t];
    const float* g = a.g[t];
    const int64_t n = a.size[t];
    const float scale = a.scal->lr * a.scal->clip_coef;
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
        p[i] = p[i] - scale * g[i];
}

extern "C" int fx_mt_sqnorm(const float* const* grads_host, const int64_t* sizes_host, int32_t n,
                            float* sq_partials, fx_stream_t stream) {
    FX_CHECK_ARG(n >= 0 && n <= FX_MT_MAX, "fx_mt_sqnorm: n=%d not in [0,%d]", n, FX_MT_MAX);
    if (n == 0) return FX_OK;
    FX_CHECK_ARG(grads_host && sizes_host && sq_partials, "fx_mt_sqnorm: null pointer");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        FX_CHECK_ARG(grads_host[i] || sizes_host[i] == 0, "fx_mt_sqnorm: null grad %d", i);
        a.g[i] = grads_host[i];
        a.size[i] = sizes_host[i];
    }
    a.sq_partials = sq_partials;
    hipLaunchKernelGGL(k_mt_sqnorm, dim3(FX_MT_BLOCKS, n), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_mt_adam(float* const* params_host, const float* const* grads_host,
                          float* const* m_host, float* const* v_host, const int64_t* sizes_host,
                          int32_t n, const fx_scalars* scal, fx_stream_t stream) {
    FX_CHECK_ARG(n >= 0 && n <= FX_MT_MAX, "fx_mt_adam: n=%d not in [0,%d]", n, FX_MT_MAX);
    if (n == 0) return FX_OK;
    FX_CHECK_ARG(params_host && grads_host && m_host && v_host && sizes_host && scal,
                 "fx_mt_adam: null pointer");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        FX_CHECK_ARG(sizes_host[i] == 0 ||
                         (params_host[i] && grads_host[i] && m_host[i] && v_host[i]),
                     "fx_mt_adam: null tensor %d", i);
        a.p[i] = params_host[i];
        a.g[i] = grads_host[i];
        a.m[i] = m_host[i];
        a.v[i] = v_host[i];
        a.size[i] = sizes_host[i];
    }
    a.scal = scal;
    hipLaunchKernelGGL(k_mt_adam, dim3(FX_MT_BLOCKS, n), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_mt_sgd(float* const* params_host, const float* const* grads_host,
                         const int64_t* sizes_host, int32_t n, const fx_scalars* scal,
                         fx_stream_t stream) {
    FX_CHECK_ARG(n >= 0 && n <= FX_MT_MAX, "fx_mt_sgd: n=%d not in [0,%d]", n, FX_MT_MAX);
    if (n == 0) return FX_OK;
    FX_CHECK_ARG(params_host && grads_host && sizes_host && scal, "fx_mt_sgd: null pointer");
    MtArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n; ++i) {
        FX_CHECK_ARG(sizes_host[i] == 0 || (params_host[i] && grads_host[i]),
                     "fx_mt_sgd: null tensor %d", i);
        a.p[i] = params_host[i];
        a.g[i] = grads_host[i];
        a.size[i] = sizes_host[i];
    }
    a.scal = scal;
    hipLaunchKernelGGL(k_mt_sgd, dim3(FX_MT_BLOCKS, n), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
