// fx_dedup_lds.hip — the in-LDS de-dup for id plans the column path cannot take (round 6; BASELINE north_star:
// "hashed index dedup in LDS"): sequence columns that alias their target's table (c4 DIN: 51 columns x 4096
// lookups of ONE 847 K-row table), shared tables, B > 8192.
//
// The generic path sorts all (row, position) pairs with a device-wide LSD radix sort: key build + 3 x (histogram +
// scatter) + heads + scan = 10 launches, 75 us of c4's 471 us step.  A de-dup does not need ascending rows — it
// needs every row's lookups side by side, in position order (the gradient sum's order), and a deterministic
// order of the unique rows.  So the rows are HASHED into 256 buckets by their low 8 bits (consecutive hot ids of
// a power-law column, and the two or three rows of a tiny vocabulary, land in different buckets — a split by
// the high bits would put c4's eight small tables, 32 K lookups, into one) by ONE stable partition pass through
// global memory, and every bucket (c4: ~ 600 pairs, padding positions never enter a bucket) is then sorted by
// ONE workgroup entirely in LDS over the remaining bits:
//   k_bk_keys_hist   keys from the id matrix (pad / bad ids -> dropped) + per-tile bucket histogram   (plain stores)
//   k_bk_partition   bucket offsets from the tile histograms, rank by wave match masks, scatter (stable)
//   k_bk_sort        256 workgroups: bucket -> LDS, stable LSD radix sort on bits 8.., head flags, block scan
//   k_bk_finish      unique rows / segment starts / unique index of every sorted lookup, sentinel tail
// 4 launches; no atomics on global memory, no memset nodes, no look-back: captures into a hipGraph, deterministic.
// Order of the result: by (row & 255, row >> 8), a row's lookups by position.
// A bucket that does not fit the LDS buffers (> 8192 pairs: one row looked up > 8192 times in a batch, or an
// adversarial id set) is sorted by the same code chunk by chunk through global memory — slow, correct.
#include <stdlib.h>

#include "fx_common.h"

namespace {

constexpr int BK_T = 1024;                // threads per workgroup (16 waves)
constexpr int BK_IPT = 8;                 // items per thread and chunk
constexpr int BK_TILE = BK_T * BK_IPT;    // 8192 items: a partition tile, and the LDS capacity of a bucket
constexpr int BK_NB = 256;                // buckets

__device__ __forceinline__ uint32_t bk_wave_incl_scan(uint32_t x, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

// rank of a lane's digit among the valid lanes of its wave that hold the same digit and come before it, on top
// of the wave's running count of that digit (hist_row: the wave's private 256-entry row in LDS; every lane of
// a match group reads the same word, the group's lowest lane bumps it).  All 64 lanes call this.
__device__ __forceinline__ uint32_t bk_wave_rank(uint32_t* hist_row, uint32_t d, bool valid, int lane) {
    uint64_t m = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        const bool bit = (d >> b) & 1u;
        const uint64_t bal = __ballot(bit);
        m &= bit ? bal : ~bal;
    }
    const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    uint32_t base = 0;
    if (valid) base = hist_row[d];
    __builtin_amdgcn_wave_barrier();
    if (valid && below == 0u) hist_row[d] = base + (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    return base + below;
}

// launch 1: keys[i] for every lookup i = b * C + c (sentinel: padding_idx / out-of-range id), bucket counts of
// the tile's valid keys -> bh[tile][256].  Block 0 also opens the optimizer step when asked to.
// T threads x 8 items: tiles of 2048 lookups (T = 256: c4's 262 K lookups fill 128 CUs) up to 2^19 lookups, of
// 8192 (T = 1024) above — the partition reads one count row per tile.
template <int T>
__global__ __launch_bounds__(T) void k_bk_keys_hist(const int32_t* __restrict__ ids, int64_t ids_ld, int64_t n,
                                                    int C, const int64_t* __restrict__ col_row_base,
                                                    const int32_t* __restrict__ col_vocab,
                                                    const int32_t* __restrict__ col_pad, uint32_t sentinel,
                                                    uint32_t* __restrict__ keys, uint32_t* __restrict__ bh,
                                                    fx_scalars* begin_scal) {
    __shared__ uint32_t hist[BK_NB];
    if (begin_scal != nullptr && blockIdx.x == 0 && threadIdx.x == 0) fx_begin_step_dev(begin_scal);
    for (int i = threadIdx.x; i < BK_NB; i += T) hist[i] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * (T * BK_IPT);
    uint32_t key[BK_IPT];
#pragma unroll
    for (int r = 0; r < BK_IPT; ++r) {
        const int64_t i = base + r * T + threadIdx.x;
        key[r] = sentinel;
        if (i < n) {
            const uint32_t b = (uint32_t)i / (uint32_t)C;          // (n < 2^31)
            const int c = (int)((uint32_t)i - b * (uint32_t)C);
            const int32_t id = ids[(int64_t)b * ids_ld + c];
            if (id >= 0 && id < col_vocab[c] && id != col_pad[c]) key[r] = (uint32_t)(col_row_base[c] + id);
        }
    }
#pragma unroll
    for (int r = 0; r < BK_IPT; ++r) {
        const int64_t i = base + r * T + threadIdx.x;
        if (i < n) {
            keys[i] = key[r];
            if (key[r] != sentinel) atomicAdd(&hist[key[r] & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < BK_NB; i += T) bh[(int64_t)blockIdx.x * BK_NB + i] = hist[i];
}

// launch 2: stable partition of the valid (key, position) pairs by key & 255.  Bucket d of tile t starts at
// sum_{d' < d} tot[d'] + sum_{t' < t} bh[t'][d]; inside the tile, wave chunks in index order.
template <int T>
__global__ __launch_bounds__(T) void k_bk_partition(const uint32_t* __restrict__ keys, int64_t n,
                                                    uint32_t sentinel, const uint32_t* __restrict__ bh,
                                                    int nblk, uint32_t* __restrict__ part_key,
                                                    uint32_t* __restrict__ part_pos,
                                                    uint32_t* __restrict__ btot) {
    constexpr int NW = T / 64, S = T / BK_NB;
    __shared__ uint32_t hist[NW][BK_NB];
    __shared__ uint32_t dbase[BK_NB];
    __shared__ uint32_t part[2][S][BK_NB];
    __shared__ uint32_t wtot[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t base = (int64_t)blockIdx.x * (T * BK_IPT);
    uint32_t kk[BK_IPT], rk[BK_IPT];
#pragma unroll
    for (int r = 0; r < BK_IPT; ++r) {
        const int64_t i = base + (w * BK_IPT + r) * 64 + lane;
        kk[r] = i < n ? keys[i] : sentinel;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) hist[w][lane + 64 * j] = 0u;
    {
        // count rows of all tiles: slice q of the threads takes tiles q, q + S, ...; the loads are independent
        const int d = threadIdx.x & (BK_NB - 1), q = threadIdx.x / BK_NB;
        uint32_t before = 0, total = 0;
#pragma unroll 8
        for (int t = q; t < nblk; t += S) {
            const uint32_t x = bh[(int64_t)t * BK_NB + d];
            total += x;
            before += t < (int)blockIdx.x ? x : 0u;
        }
        part[0][q][d] = before;
        part[1][q][d] = total;
    }
    __syncthreads();
    if (threadIdx.x < BK_NB) {
        const int d = threadIdx.x;
        uint32_t before = 0, total = 0;
#pragma unroll
        for (int q = 0; q < S; ++q) {
            before += part[0][q][d];
            total += part[1][q][d];
        }
        const uint32_t inc = bk_wave_incl_scan(total, lane);
        if (lane == 63) wtot[w] = inc;
        dbase[d] = inc - total + before;
        if (blockIdx.x == 0) btot[d] = total;
    }
#pragma unroll
    for (int r = 0; r < BK_IPT; ++r) rk[r] = bk_wave_rank(hist[w], kk[r] & 255u, kk[r] != sentinel, lane);
    __syncthreads();
    if (threadIdx.x < BK_NB) {
        const int d = threadIdx.x;
        uint32_t add = 0;
        for (int ww = 0; ww < w; ++ww) add += wtot[ww];
        dbase[d] += add;
        uint32_t run = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) {
            const uint32_t t = hist[ww][d];
            hist[ww][d] = run;
            run += t;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < BK_IPT; ++r) {
        if (kk[r] != sentinel) {
            const uint32_t d = kk[r] & 255u;
            const uint32_t dst = dbase[d] + hist[w][d] + rk[r];
            part_key[dst] = kk[r];
            part_pos[dst] = (uint32_t)(base + (w * BK_IPT + r) * 64 + lane);
        }
    }
}

// launch 3: one workgroup per bucket.  FITS: the bucket lives in two LDS buffers; otherwise the same passes run
// chunk by chunk between the partition arrays (buffer 0) and the output arrays (buffer 1) in global memory.
struct BkSortArgs {
    uint32_t* part_key;       // buffer 0 (the partition's output)
    uint32_t* part_pos;
    uint32_t* sorted_key;     // buffer 1 = the result
    uint32_t* sorted_pos;
    uint32_t* col_scan;       // running unique count inside the bucket (inclusive), per sorted pair
    uint32_t* col_cnt;        // unique rows of bucket d
    const uint32_t* btot;     // pairs of bucket d
    int end_bit;              // keys use bits [0, end_bit)
};

template <bool FITS>
__device__ __forceinline__ void bk_sort_bucket(const BkSortArgs& a, uint32_t lo, uint32_t cnt, int d,
                                               uint32_t (*kbuf)[BK_TILE], uint32_t (*pbuf)[BK_TILE],
                                               uint32_t (*hist)[BK_NB], uint32_t* dbase, uint32_t* wtot) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    uint32_t* gk[2] = {a.part_key + lo, a.sorted_key + lo};
    uint32_t* gp[2] = {a.part_pos + lo, a.sorted_pos + lo};
    // rounds a wave runs per chunk: a small bucket occupies the first `ipt` rounds of every wave
    const int ipt = FITS ? (int)((cnt + BK_T - 1) / BK_T) : BK_IPT;
    const uint32_t chunk = (uint32_t)ipt * BK_T;
    if (FITS) {
        for (uint32_t i = threadIdx.x; i < chunk; i += BK_T) {
            kbuf[0][i] = i < cnt ? gk[0][i] : 0xFFFFFFFFu;       // fill: every digit 255, after the real pairs
            pbuf[0][i] = i < cnt ? gp[0][i] : 0xFFFFFFFFu;
        }
        __syncthreads();
    }
    int src = 0;
    for (int shift = 8; shift < a.end_bit; shift += 8) {
        // digit totals of the whole bucket -> exclusive offsets
        if (!FITS) {
            if (threadIdx.x < BK_NB) dbase[threadIdx.x] = 0u;
            __syncthreads();
            for (uint32_t i = threadIdx.x; i < cnt; i += BK_T) atomicAdd(&dbase[(gk[src][i] >> shift) & 255u], 1u);
            __syncthreads();
            if (threadIdx.x < BK_NB) {
                const uint32_t t = dbase[threadIdx.x];
                const uint32_t inc = bk_wave_incl_scan(t, lane);
                if (lane == 63) wtot[w] = inc;
                dbase[threadIdx.x] = inc - t;
            }
            __syncthreads();
            if (threadIdx.x < BK_NB) {
                uint32_t add = 0;
                for (int ww = 0; ww < w; ++ww) add += wtot[ww];
                dbase[threadIdx.x] += add;
            }
            __syncthreads();
        }
        for (uint32_t c0 = 0; c0 < (FITS ? 1u : cnt); c0 += chunk) {
#pragma unroll
            for (int j = 0; j < 4; ++j) hist[w][lane + 64 * j] = 0u;
            uint32_t kk[BK_IPT], pp[BK_IPT], rk[BK_IPT];
#pragma unroll
            for (int r = 0; r < BK_IPT; ++r) {
                kk[r] = 0xFFFFFFFFu;
                pp[r] = 0xFFFFFFFFu;
                rk[r] = 0u;
                if (r < ipt) {
                    const uint32_t i = c0 + (uint32_t)(w * ipt + r) * 64u + lane;
                    bool valid;
                    if (FITS) {
                        kk[r] = kbuf[src][i];
                        pp[r] = pbuf[src][i];
                        valid = true;                       // (fill pairs take part: they sort to the tail)
                    } else {
                        valid = i < cnt;
                        if (valid) {
                            kk[r] = gk[src][i];
                            pp[r] = gp[src][i];
                        }
                    }
                    rk[r] = bk_wave_rank(hist[w], (kk[r] >> shift) & 255u, valid, lane);
                }
            }
            __syncthreads();
            uint32_t ctot = 0;
            if (threadIdx.x < BK_NB) {
                const int dd = threadIdx.x;
                uint32_t run = 0;
#pragma unroll
                for (int ww = 0; ww < 16; ++ww) {
                    const uint32_t t = hist[ww][dd];
                    hist[ww][dd] = run;
                    run += t;
                }
                ctot = run;
                if (FITS) {                                   // one chunk: its totals are the bucket's
                    const uint32_t inc = bk_wave_incl_scan(run, lane);
                    if (lane == 63) wtot[w] = inc;
                    dbase[dd] = inc - run;
                }
            }
            __syncthreads();
            if (FITS && threadIdx.x < BK_NB) {
                uint32_t add = 0;
                for (int ww = 0; ww < w; ++ww) add += wtot[ww];
                dbase[threadIdx.x] += add;
            }
            if (FITS) __syncthreads();
#pragma unroll
            for (int r = 0; r < BK_IPT; ++r) {
                if (r < ipt) {
                    const uint32_t i = c0 + (uint32_t)(w * ipt + r) * 64u + lane;
                    const uint32_t dd = (kk[r] >> shift) & 255u;
                    if (FITS) {
                        const uint32_t dst = dbase[dd] + hist[w][dd] + rk[r];
                        kbuf[src ^ 1][dst] = kk[r];
                        pbuf[src ^ 1][dst] = pp[r];
                    } else if (i < cnt) {
                        const uint32_t dst = dbase[dd] + hist[w][dd] + rk[r];
                        gk[src ^ 1][dst] = kk[r];
                        gp[src ^ 1][dst] = pp[r];
                    }
                }
            }
            __syncthreads();
            if (!FITS) {
                if (threadIdx.x < BK_NB) dbase[threadIdx.x] += ctot;
                __syncthreads();
            }
        }
        if (!FITS) __threadfence_block();
        src ^= 1;
    }
    // head flags (thread t owns `ipt` consecutive pairs of a chunk), block scan, the bucket leaves for the result
    uint32_t carry = 0;
    for (uint32_t c0 = 0; c0 < (FITS ? 1u : cnt); c0 += chunk) {
        uint32_t k[BK_IPT], p[BK_IPT], flag[BK_IPT], h = 0;
        const uint32_t i0 = c0 + threadIdx.x * (uint32_t)ipt;
        uint32_t prev = 0xFFFFFFFFu;
        if (i0 > 0 && i0 < cnt) prev = FITS ? kbuf[src][i0 - 1] : gk[src][i0 - 1];
#pragma unroll
        for (int j = 0; j < BK_IPT; ++j) {
            k[j] = 0xFFFFFFFFu;
            p[j] = 0xFFFFFFFFu;
            flag[j] = 0u;
            if (j < ipt && i0 + j < cnt) {
                k[j] = FITS ? kbuf[src][i0 + j] : gk[src][i0 + j];
                p[j] = FITS ? pbuf[src][i0 + j] : gp[src][i0 + j];
                flag[j] = (i0 + j == 0 || k[j] != prev) ? 1u : 0u;
                prev = k[j];
                h += flag[j];
            }
        }
        if (!FITS) __syncthreads();                 // (every read of the source chunk before any write below)
        const uint32_t inc = bk_wave_incl_scan(h, lane);
        if (lane == 63) hist[0][w] = inc;           // (the histogram rows are free now)
        __syncthreads();
        uint32_t before = carry + inc - h, total = 0;
        for (int ww = 0; ww < 16; ++ww) {
            const uint32_t t = hist[0][ww];
            if (ww < w) before += t;
            total += t;
        }
#pragma unroll
        for (int j = 0; j < BK_IPT; ++j) {
            if (j < ipt && i0 + j < cnt) {
                before += flag[j];
                a.sorted_key[lo + i0 + j] = k[j];
                a.sorted_pos[lo + i0 + j] = p[j];
                a.col_scan[lo + i0 + j] = before;
            }
        }
        carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) a.col_cnt[d] = carry;
}

__global__ __launch_bounds__(BK_T) void k_bk_sort(BkSortArgs a) {
    __shared__ uint32_t kbuf[2][BK_TILE];
    __shared__ uint32_t pbuf[2][BK_TILE];
    __shared__ uint32_t hist[16][BK_NB];
    __shared__ uint32_t dbase[BK_NB];
    __shared__ uint32_t wtot[16];
    const int d = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // first pair of the bucket = pairs of the buckets before it
    uint32_t x = 0;
    if (threadIdx.x < BK_NB && (int)threadIdx.x < d) x = a.btot[threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o, 64);
    if (lane == 0) wtot[w] = x;
    __syncthreads();
    const uint32_t lo = (wtot[0] + wtot[1]) + (wtot[2] + wtot[3]);
    const uint32_t cnt = a.btot[d];
    __syncthreads();
    if (cnt == 0u) {
        if (threadIdx.x == 0) a.col_cnt[d] = 0u;
        return;
    }
    if (cnt <= (uint32_t)BK_TILE) bk_sort_bucket<true>(a, lo, cnt, d, kbuf, pbuf, hist, dbase, wtot);
    else bk_sort_bucket<false>(a, lo, cnt, d, kbuf, pbuf, hist, dbase, wtot);
}

// launch 4: unique index of every sorted pair, unique rows, segment starts; the sentinel tail
__global__ __launch_bounds__(256) void k_bk_finish(uint32_t* __restrict__ sorted_key, uint32_t* __restrict__ sorted_pos,
                                                   const uint32_t* __restrict__ col_scan,
                                                   const uint32_t* __restrict__ col_cnt,
                                                   const uint32_t* __restrict__ btot, int64_t n, uint32_t sentinel,
                                                   uint32_t* __restrict__ uniq_row, uint32_t* __restrict__ seg_start,
                                                   int32_t* __restrict__ n_unique, uint32_t* __restrict__ sorted_uid) {
    __shared__ uint32_t ioff[BK_NB + 1], uoff[BK_NB + 1], ws[2][4];
    {
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        const uint32_t bi = btot[threadIdx.x], bu = col_cnt[threadIdx.x];
        const uint32_t ii = bk_wave_incl_scan(bi, lane), iu = bk_wave_incl_scan(bu, lane);
        if (lane == 63) {
            ws[0][w] = ii;
            ws[1][w] = iu;
        }
        __syncthreads();
        uint32_t ai = 0, au = 0;
        for (int ww = 0; ww < w; ++ww) {
            ai += ws[0][ww];
            au += ws[1][ww];
        }
        ioff[threadIdx.x + 1] = ai + ii;
        uoff[threadIdx.x + 1] = au + iu;
        if (threadIdx.x == 0) ioff[0] = uoff[0] = 0u;
        __syncthreads();
    }
    const int64_t n_valid = ioff[BK_NB];
    if (n_valid == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        *n_unique = 0;
        seg_start[0] = 0u;
    }
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        if (i >= n_valid) {
            sorted_key[i] = sentinel;
            sorted_pos[i] = 0xFFFFFFFFu;
            if (sorted_uid) sorted_uid[i] = 0xFFFFFFFFu;
            continue;
        }
        int lo = 0, hi = BK_NB;                         // bucket c: ioff[c] <= i < ioff[c + 1]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ioff[mid] <= (uint32_t)i) lo = mid; else hi = mid;
        }
        const uint32_t k = sorted_key[i];
        const uint32_t u = uoff[lo] + col_scan[i];
        const bool head = (i == 0) || (sorted_key[i - 1] != k);
        if (sorted_uid) sorted_uid[i] = u - 1;
        if (head) {
            uniq_row[u - 1] = k;
            seg_start[u - 1] = (uint32_t)i;
        }
        if (i == n_valid - 1) {
            seg_start[u] = (uint32_t)n_valid;
            *n_unique = (int32_t)u;
        }
    }
}

}  // namespace

static inline int bk_part_tile(int64_t n) { return n <= (int64_t)256 * 2048 ? 2048 : BK_TILE; }   // <= 256 tiles

// workspace words: keys / col_scan [n] | part_key [n] | part_pos [n] | bh [nblk x 256] | btot [256] | col_cnt [256]
size_t fx_dedup_buckets_bytes(int64_t n) {
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) / 256 * 256;
    const size_t nblk = (size_t)fx_ceil_div(n, bk_part_tile(n));
    return 3 * arr + (nblk + 2) * BK_NB * sizeof(uint32_t) + 256;
}

bool fx_dedup_buckets_ok(int64_t n) {
    static const bool on = []() {
        const char* e = getenv("FX_DEDUP_BUCKETS");
        return !(e && atoi(e) == 0);
    }();
    // 256 buckets x 8192 pairs in LDS; the partition reads nblk x 256 tile counts per workgroup
    return on && n > 0 && n <= (int64_t)BK_NB * BK_TILE;
}

int fx_dedup_buckets_launch(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                            const int64_t* col_row_base, const int32_t* col_vocab, const int32_t* col_pad,
                            uint32_t sentinel, void* workspace, uint32_t* sorted_key, uint32_t* sorted_pos,
                            uint32_t* uniq_row, uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                            fx_scalars* begin_scal, hipStream_t s) {
    const int64_t n = B * (int64_t)C;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) / 256 * 256;
    const int tile = bk_part_tile(n);
    const int nblk = (int)fx_ceil_div(n, tile);
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* keys = reinterpret_cast<uint32_t*>(w);             // col_scan once the partition has read it
    uint32_t* part_key = reinterpret_cast<uint32_t*>(w + arr);
    uint32_t* part_pos = reinterpret_cast<uint32_t*>(w + 2 * arr);
    uint32_t* bh = reinterpret_cast<uint32_t*>(w + 3 * arr);
    uint32_t* btot = bh + (size_t)nblk * BK_NB;
    uint32_t* col_cnt = btot + BK_NB;
    if (tile == 2048) {
        hipLaunchKernelGGL(k_bk_keys_hist<256>, dim3(nblk), dim3(256), 0, s, ids, ids_ld, n, (int)C, col_row_base,
                           col_vocab, col_pad, sentinel, keys, bh, begin_scal);
        hipLaunchKernelGGL(k_bk_partition<256>, dim3(nblk), dim3(256), 0, s, keys, n, sentinel, bh, nblk,
                           part_key, part_pos, btot);
    } else {
        hipLaunchKernelGGL(k_bk_keys_hist<1024>, dim3(nblk), dim3(1024), 0, s, ids, ids_ld, n, (int)C,
                           col_row_base, col_vocab, col_pad, sentinel, keys, bh, begin_scal);
        hipLaunchKernelGGL(k_bk_partition<1024>, dim3(nblk), dim3(1024), 0, s, keys, n, sentinel, bh, nblk,
                           part_key, part_pos, btot);
    }
    BkSortArgs a;
    a.part_key = part_key;
    a.part_pos = part_pos;
    a.sorted_key = sorted_key;
    a.sorted_pos = sorted_pos;
    a.col_scan = keys;
    a.col_cnt = col_cnt;
    a.btot = btot;
    a.end_bit = 1;
    while (a.end_bit < 32 && (sentinel >> a.end_bit) != 0u) ++a.end_bit;
    hipLaunchKernelGGL(k_bk_sort, dim3(BK_NB), dim3(BK_T), 0, s, a);
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 256 * 8) blocks = 256 * 8;
    hipLaunchKernelGGL(k_bk_finish, dim3((unsigned)blocks), dim3(256), 0, s, sorted_key, sorted_pos, keys, col_cnt,
                       btot, n, sentinel, uniq_row, seg_start, n_unique, sorted_uid);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
