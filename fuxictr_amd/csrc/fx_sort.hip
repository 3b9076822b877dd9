// fx_sort.hip — device-wide STABLE sort of (uint32 key, uint32 value) pairs for the generic de-dup
// (sequence columns / shared tables, fx_sparse.hip) and the AUC rank pass (fx_metrics.hip).
//
// Why not the library: rocPRIM's device radix sort picks its merge sort below 1 M items (c4's
// 262 K lookups: a block sort + 16 merge launches + a copy-back, ~200 us of a 0.87 ms DIN step,
// profiles/r02_step_timeline_din_before.txt) and its Onesweep path — forced with MergeSortLimit=0 —
// faults on this stack as soon as it runs on a stream other than the null stream or inside a
// captured hipGraph (measured: HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION at 16 K and 262 K items),
// which is where the training step runs.
//
// The sort here is a least-significant-digit radix sort, 8 bits a pass, over only the key bits in
// use (c4's 2.8 M packed rows: 22 bits = 3 passes).  A pass is two launches and nothing else (no
// memset nodes, no look-back spinning, no cross-workgroup fences), so it captures into a hipGraph:
//   k_rs_hist     every workgroup counts the digits of its 2048-item tile in LDS and publishes
//                 bh[tile][digit] (plain stores) plus integer atomics into gh[tile / G][digit]
//                 and tot[digit] (integer sums: the result does not depend on arrival order);
//   k_rs_scatter  every workgroup rebuilds the global offset of (digit, its tile) from tot / gh /
//                 bh (a 256-wide scan + <= ng + G - 1 coalesced loads per thread), ranks its items
//                 by digit with wave-wide match masks (8 ballots an item, in tile order, hence
//                 stable) and scatters them.
// HBM-bound integer work: per pass every pair is read twice and written once (20 B an item); at
// c4's size the launches are latency-bound (~5 us each), not bandwidth-bound.
#include "fx_common.h"

namespace {

constexpr int RS_T = 256;                 // threads per workgroup (4 waves)
constexpr int RS_IPT = 8;                 // items per thread
constexpr int RS_TILE = RS_T * RS_IPT;    // 2048 items per workgroup

// item r of this thread: tile order = wave-major, then round, then lane (coalesced 256 B rounds)
__device__ __forceinline__ int64_t rs_index(int64_t tile_base, int r) {
    return tile_base + (int64_t)(threadIdx.x >> 6) * (64 * RS_IPT) + r * 64 + (threadIdx.x & 63);
}

__global__ __launch_bounds__(RS_T) void k_rs_zero(uint32_t* p, int64_t words) {
    for (int64_t i = (int64_t)blockIdx.x * RS_T + threadIdx.x; i < words;
         i += (int64_t)gridDim.x * RS_T)
        p[i] = 0u;
}

__global__ __launch_bounds__(RS_T) void k_rs_hist(const uint32_t* __restrict__ keys, int64_t n,
                                                  int shift, uint32_t* __restrict__ bh,
                                                  uint32_t* __restrict__ gh, int group,
                                                  uint32_t* __restrict__ tot) {
    __shared__ uint32_t hist[256];
    hist[threadIdx.x] = 0u;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const int64_t i = rs_index(base, r);
        if (i < n) atomicAdd(&hist[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    const uint32_t h = hist[threadIdx.x];
    bh[(int64_t)blockIdx.x * 256 + threadIdx.x] = h;
    if (h) {
        atomicAdd(&gh[(blockIdx.x / group) * 256 + threadIdx.x], h);
        atomicAdd(&tot[threadIdx.x], h);
    }
}

template <bool IDENTITY>
__global__ __launch_bounds__(RS_T) void k_rs_scatter(
    const uint32_t* __restrict__ keys, const uint32_t* __restrict__ vals, int64_t n, int shift,
    const uint32_t* __restrict__ bh, const uint32_t* __restrict__ gh, int group,
    const uint32_t* __restrict__ tot, uint32_t* __restrict__ keys_out,
    uint32_t* __restrict__ vals_out, uint32_t* __restrict__ zero_next, int zero_words) {
    __shared__ uint32_t dbase[256];          // global offset of (digit, this tile)
    // per-wave digit counts -> exclusive prefix over waves.  volatile: within a wave the lanes of
    // one digit group all read the counter, then the group's first lane bumps it (lockstep order)
    __shared__ volatile uint32_t wcnt[4][256];
    __shared__ uint32_t wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // the counters of the NEXT pass are zeroed here (nobody touches them before that pass's
    // k_rs_hist, which is launched after this kernel)
    if (zero_next)
        for (int i = blockIdx.x * RS_T + tid; i < zero_words; i += gridDim.x * RS_T) zero_next[i] = 0u;

    // ---- global offset of every digit for this tile ------------------------------------
    {
        const uint32_t t = tot[tid];
        uint32_t incl = t;                    // inclusive scan over the 256 digits: waves, then block
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o, 64);
            if (lane >= o) incl += up;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        uint32_t off = before + incl - t;
        const int g = blockIdx.x / group;
        for (int q = 0; q < g; ++q) off += gh[q * 256 + tid];
        for (int b = g * group; b < (int)blockIdx.x; ++b) off += bh[(int64_t)b * 256 + tid];
        dbase[tid] = off;
    }
#pragma unroll
    for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0u;
    __syncthreads();

    // ---- rank inside the tile: per wave, round by round ------------------------------------
    const int64_t base = (int64_t)blockIdx.x * RS_TILE;
    const unsigned long long below = (1ull << lane) - 1ull;
    uint32_t k[RS_IPT], v[RS_IPT], local[RS_IPT];
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const int64_t i = rs_index(base, r);
        k[r] = 0u;
        v[r] = 0u;
        if (i < n) {
            k[r] = keys[i];
            v[r] = IDENTITY ? (uint32_t)i : vals[i];
        }
    }
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        const bool valid = rs_index(base, r) < n;
        const uint32_t d = (k[r] >> shift) & 255u;
        unsigned long long mask = __ballot(valid);
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) {
            const unsigned long long bal = __ballot((d >> bit) & 1u);
            mask &= ((d >> bit) & 1u) ? bal : ~bal;
        }
        local[r] = 0u;
        if (valid) {
            const uint32_t prev = wcnt[wave][d];               // every lane of the group reads ...
            local[r] = prev + (uint32_t)__popcll(mask & below);
            if ((mask & below) == 0ull)                        // ... then its first lane adds
                wcnt[wave][d] = prev + (uint32_t)__popcll(mask);
        }
    }
    __syncthreads();
    {
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const uint32_t c = wcnt[w][tid];
            wcnt[w][tid] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RS_IPT; ++r) {
        if (rs_index(base, r) < n) {
            const uint32_t d = (k[r] >> shift) & 255u;
            const uint32_t dst = dbase[d] + wcnt[wave][d] + local[r];
            keys_out[dst] = k[r];
            vals_out[dst] = v[r];
        }
    }
}

struct RsLayout {
    int nblk, group, ng;
    size_t counter_words;      // one set: gh [ng x 256] + tot [256]
    size_t bh_words;           // [nblk x 256]
};

inline RsLayout rs_layout(int64_t n) {
    RsLayout L;
    L.nblk = (int)fx_ceil_div(n, RS_TILE);
    L.group = 32;              // ~sqrt(nblk): the offset rebuild reads <= ng + group rows a tile
    while ((int64_t)L.group * L.group < L.nblk) L.group *= 2;
    L.ng = (int)fx_ceil_div(L.nblk, L.group);
    L.counter_words = (size_t)256 * (L.ng + 1);
    L.bh_words = (size_t)256 * L.nblk;
    return L;
}

}  // namespace

// temp layout: [counters set 0][counters set 1][bh]
size_t fx_sort_temp_bytes(int64_t n) {
    if (n <= 0) return 256;
    const RsLayout L = rs_layout(n);
    return (2 * L.counter_words + L.bh_words) * sizeof(uint32_t) + 256;
}

size_t fx_sort_zero_words(int64_t n) { return n <= 0 ? 0 : rs_layout(n).counter_words; }

// keys_in/vals_in are not modified.  vals_in == nullptr sorts (key, index).  The result lands in
// keys_out/vals_out; keys_tmp/vals_tmp are n-word scratch arrays (unused when one pass suffices).
// `zeroed` says the caller's previous kernel on this stream already cleared the first
// fx_sort_zero_words(n) words of temp (saves one launch).
int fx_sort_pairs_u32(const uint32_t* keys_in, const uint32_t* vals_in, uint32_t* keys_out,
                      uint32_t* vals_out, uint32_t* keys_tmp, uint32_t* vals_tmp, int64_t n,
                      unsigned end_bit, void* temp, bool zeroed, hipStream_t s) {
    if (n <= 0) return FX_OK;
    FX_CHECK_ARG(n < (int64_t)0x7FFFFFFF, "fx_sort_pairs_u32: too many items (%lld)", (long long)n);
    FX_CHECK_ARG(end_bit >= 1 && end_bit <= 32, "fx_sort_pairs_u32: end_bit=%u", end_bit);
    const RsLayout L = rs_layout(n);
    uint32_t* cnt[2] = {reinterpret_cast<uint32_t*>(temp),
                        reinterpret_cast<uint32_t*>(temp) + L.counter_words};
    uint32_t* bh = cnt[1] + L.counter_words;
    const int passes = (int)((end_bit + 7) / 8);
    if (!zeroed) {
        hipLaunchKernelGGL(k_rs_zero, dim3(8), dim3(RS_T), 0, s, cnt[0], (int64_t)L.counter_words);
        FX_CHECK_LAUNCH();
    }
    const uint32_t* src_k = keys_in;
    const uint32_t* src_v = vals_in;
    for (int p = 0; p < passes; ++p) {
        const bool to_out = ((passes - 1 - p) % 2) == 0;
        uint32_t* dst_k = to_out ? keys_out : keys_tmp;
        uint32_t* dst_v = to_out ? vals_out : vals_tmp;
        uint32_t* gh = cnt[p & 1];
        uint32_t* tot = gh + (size_t)256 * L.ng;
        uint32_t* next = (p + 1 < passes) ? cnt[(p + 1) & 1] : nullptr;
        hipLaunchKernelGGL(k_rs_hist, dim3(L.nblk), dim3(RS_T), 0, s, src_k, n, 8 * p, bh, gh,
                           L.group, tot);
        if (p == 0 && vals_in == nullptr)
            hipLaunchKernelGGL(k_rs_scatter<true>, dim3(L.nblk), dim3(RS_T), 0, s, src_k, src_v, n,
                               8 * p, bh, gh, L.group, tot, dst_k, dst_v, next,
                               (int)L.counter_words);
        else
            hipLaunchKernelGGL(k_rs_scatter<false>, dim3(L.nblk), dim3(RS_T), 0, s, src_k, src_v, n,
                               8 * p, bh, gh, L.group, tot, dst_k, dst_v, next,
                               (int)L.counter_words);
        FX_CHECK_LAUNCH();
        src_k = dst_k;
        src_v = dst_v;
    }
    return FX_OK;
}
