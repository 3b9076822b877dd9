// fx_gemm.hip — fp32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32 products,
// fp32 accumulate, 157 TFLOP/s dense peak on MI355X) with a fused epilogue, plus the column-sum
// (bias gradient) and sigmoid+BCE kernels of the dense tower.
//
// Replaces the aten::addmm / relu / mul / add launches of
//   fuxictr/pytorch/layers/blocks/mlp_block.py:96            (Linear -> ReLU stack)
//   fuxictr/pytorch/layers/interactions/cross_net.py:126-129 (X_{i+1} = X_i + X_0 * (W X_i + b))
// and their autograd (dX = dZ W, dW = dZ^T X, db = colsum dZ) triggered at rank_model.py:320.
//
// Tiling (one wave = 64 lanes, 4 waves per workgroup, one workgroup per CU at B=4096):
//   block tile 128x128x32, LDS double-buffered (67.5 KB), k-major tiles T[k][m] so that an MFMA operand
//   fragment (lane l: row l&31, k = l>>5) is one conflict-free ds_read_b32;
//   wave tile 64x64 = 2x2 MFMA tiles of 32x32 -> 4 independent accumulators (64 VGPRs);
//   global->register prefetch of tile t+1 is issued before the MFMAs of tile t;
//   blockIdx is remapped so the 8 n-tiles that share one A row-panel run on the same XCD (L2).
#include "fx_common.h"
#include "fx_gemm_int.h"

#include <stdlib.h>

#include <mutex>

// Operand tile loader for an R x 32 tile (R = 64 or 128 rows of the non-contracted dimension).
// KC: element (r,k) at P[r*ld + k] (k contiguous) else at P[k*ld + r] (r contiguous).
// LDS image is always k-major T[k][LD]: LD = R+1 when filled by transposing 4-byte writes
// (conflict-free), R+4 when filled by 16-byte writes (keeps 16-B alignment).
template <int R, bool KC, bool VEC>
struct TileLoader {
    static constexpr int NST = R / 32;            // float4 staging registers per thread
    static constexpr int LD = KC ? R + 1 : R + 4;
    float4 st[NST];

    __device__ __forceinline__ void load(const float* __restrict__ P, int64_t ld, int64_t r0,
                                         int64_t Rext, int64_t k0, int64_t kend) {
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            const int q = threadIdx.x + 256 * p;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (KC) {
                const int64_t r = r0 + (q >> 3);
                const int64_t k = k0 + ((q & 7) << 2);
                if (r < Rext) {
                    const float* src = P + r * ld + k;
                    if constexpr (VEC) {
                        if (k < kend) v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (k + 0 < kend) v.x = src[0];
                        if (k + 1 < kend) v.y = src[1];
                        if (k + 2 < kend) v.z = src[2];
                        if (k + 3 < kend) v.w = src[3];
                    }
                }
            } else {
                const int64_t k = k0 + q / (R / 4);
                const int64_t r = r0 + ((q % (R / 4)) << 2);
                if (k < kend) {
                    const float* src = P + k * ld + r;
                    if constexpr (VEC) {
                        if (r < Rext) v = *reinterpret_cast<const float4*>(src);
                    } else {
                        if (r + 0 < Rext) v.x = src[0];
                        if (r + 1 < Rext) v.y = src[1];
                        if (r + 2 < Rext) v.z = src[2];
                        if (r + 3 < Rext) v.w = src[3];
                    }
                }
            }
            st[p] = v;
        }
    }

    __device__ __forceinline__ void store(float* __restrict__ T) const {
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            const int q = threadIdx.x + 256 * p;
            if constexpr (KC) {
                const int r = q >> 3, kq = (q & 7) << 2;
                T[(kq + 0) * LD + r] = st[p].x;
                T[(kq + 1) * LD + r] = st[p].y;
                T[(kq + 2) * LD + r] = st[p].z;
                T[(kq + 3) * LD + r] = st[p].w;
            } else {
                const int k = q / (R / 4), r = (q % (R / 4)) << 2;
                *reinterpret_cast<float4*>(T + k * LD + r) = st[p];
            }
        }
    }
};

// BM x BN x 32 block tile, 4 waves as 2 (m) x 2 (n); a wave owns (BM/2) x (BN/2) = MI x NJ MFMA
// tiles of 32x32.  128x128 (one workgroup per CU at 67.5 KB LDS... two fit) is the efficient
// shape when the grid has >= 2 workgroups per CU; at B = 4096 the towers give exactly 256 such
// tiles, so 128x64 / 64x64 are used there to keep 2-4 workgroups per CU in flight: the barrier /
// LDS-refill bubble of one workgroup is then covered by the MFMAs of another.
template <int BM, int BN, bool A_KC, bool B_KC, bool A_VEC, bool B_VEC>
__global__ __launch_bounds__(256) void k_gemm_f32(GemmArgs a) {
    using LoaderA = TileLoader<BM, A_KC, A_VEC>;
    using LoaderB = TileLoader<BN, B_KC, B_VEC>;
    constexpr int LDA = LoaderA::LD, LDB = LoaderB::LD;
    constexpr int MI = BM / 64, NJ = BN / 64;
    __shared__ __attribute__((aligned(16))) float As[2][FX_BK * LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][FX_BK * LDB];

    // XCD-aware tile mapping: workgroup L runs on XCD L % 8; give each XCD a contiguous range of
    // tiles (row-major over (tm, tn)) so the n-tiles sharing an A panel share one L2.
    const int64_t nwg = (int64_t)a.tiles_m * a.tiles_n;
    const int64_t L = blockIdx.x;
    int64_t T = L;
    if (nwg >= 8) {
        const int64_t q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int64_t m0 = (T / a.tiles_n) * BM;
    const int64_t n0 = (T % a.tiles_n) * BN;
    const int z = blockIdx.y;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;

    // optional fused row sums of op(A) (bias gradient when op(A) = dZ^T): blocks of the first
    // tile column add up their A tiles straight from LDS
    const bool do_rowsum = (a.epi.rowsum != nullptr) && (n0 == 0);
    float rsum = 0.f;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    LoaderA la;
    LoaderB lb;
    const int64_t nk = (kend > kbeg) ? (kend - kbeg + FX_BK - 1) / FX_BK : 0;
    if (nk > 0) {
        la.load(a.A, a.lda, m0, a.M, kbeg, kend);
        lb.load(a.B, a.ldb, n0, a.N, kbeg, kend);
        la.store(As[0]);
        lb.store(Bs[0]);
    }
    __syncthreads();
    for (int64_t t = 0; t < nk; ++t) {
        const int cur = (int)(t & 1);
        if (t + 1 < nk) {
            la.load(a.A, a.lda, m0, a.M, kbeg + (t + 1) * FX_BK, kend);
            lb.load(a.B, a.ldb, n0, a.N, kbeg + (t + 1) * FX_BK, kend);
        }
        const float* as = As[cur] + half * LDA + wm * (BM / 2) + l31;
        const float* bs = Bs[cur] + half * LDB + wn * (BN / 2) + l31;
        if (do_rowsum && threadIdx.x < BM) {
            const float* col = As[cur] + threadIdx.x;
#pragma unroll
            for (int k = 0; k < FX_BK; ++k) rsum += col[k * LDA];
        }
        // Software-pipelined fragment reads, two k-pairs deep: the LDS reads of k-pair s+2 are
        // issued right after the MFMAs of k-pair s (same register set), so an LDS latency is
        // always covered by MFMAs.  Pinned with sched_group_barrier — left alone, hipcc sinks
        // every read next to its use and pays a full LDS latency per MFMA group.
        float fa[2][MI], fb[2][NJ];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[s2][i] = as[(2 * s2) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[s2][j] = bs[(2 * s2) * LDB + 32 * j];
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
        for (int s2 = 0; s2 < FX_BK / 2; ++s2) {
            const int c = s2 & 1;
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j],
                                                                     acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, MI * NJ, 0);
            if (s2 + 2 < FX_BK / 2) {
#pragma unroll
                for (int i = 0; i < MI; ++i) fa[c][i] = as[(2 * s2 + 4) * LDA + 32 * i];
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[c][j] = bs[(2 * s2 + 4) * LDB + 32 * j];
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        if (t + 1 < nk) {
            la.store(As[cur ^ 1]);
            lb.store(Bs[cur ^ 1]);
        }
        __syncthreads();
    }

    if (do_rowsum && threadIdx.x < BM && m0 + threadIdx.x < a.M) {
        if (a.split_k > 1) a.ws[(int64_t)a.split_k * a.M * a.N + (int64_t)z * a.M + m0 + threadIdx.x] = rsum;
        else a.epi.rowsum[m0 + threadIdx.x] = rsum;
    }
    // C/D layout of 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int64_t n = n0 + wn * (BN / 2) + j * 32 + l31;
            if (n >= a.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (m >= a.M) continue;
                if (a.split_k > 1) {
                    a.ws[((int64_t)z * a.M + m) * a.N + n] = acc[i][j][r];
                } else {
                    a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc[i][j][r], m, n);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined variant for 16-byte-aligned operands (every tower GEMM of the B=4096 step).
// The kernel above stops its MFMA stream at every k-tile boundary (wait for the prefetched
// registers, 8-32 ds_writes, barrier, fragment-read latency) — ~15-20 % of a tile when only one
// workgroup fits a CU.  Here the boundary work is spread over the MFMA stream instead:
//   * two register staging sets: the global loads of tile t+2 are issued at the top of tile t,
//     the registers of tile t+1 (loaded a whole tile earlier) go to the other LDS stage during
//     MFMA groups 2..12, one ds_write after each MFMA;
//   * ONE barrier per tile after group 13; groups 14/15 already read the first fragments of tile
//     t+1 from the other stage, so the next tile starts with its MFMAs;
//   * MFMA operands in VGPR form (amdgpu_waves_per_eu(2,2)): with AGPR accumulators the compiler
//     copied all 64 of them in and out around the loop's branches.
// Out-of-range rows / the K tail are clamped addresses + zero selects (no exec-mask branches).
// ---------------------------------------------------------------------------------------------
template <int R, bool KC>
struct PipeLoader {
    static constexpr int NST = R / 32;
    static constexpr int LD = KC ? R + 1 : R + 4;
    const float* P;
    int32_t ld;
    int32_t rc[NST];       // KC: clamped row * ld ; else: clamped first row of the float4
    int32_t kl[NST];       // k of this thread's float4 inside a tile
    uint32_t rok;          // bit p: the row(s) of float4 p exist
    uint32_t voff[NST];    // byte offset of float4 p in tile 0 (valid when the rows exist)
    int32_t kbeg, kend;

    __device__ __forceinline__ void init(const float* P_, int64_t ld_, int64_t r0, int64_t Rext,
                                         int64_t kbeg_, int64_t kend_) {
        P = P_;
        ld = (int32_t)ld_;
        kbeg = (int32_t)kbeg_;
        kend = (int32_t)kend_;
        rok = 0;
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            const int q = threadIdx.x + 256 * p;
            if constexpr (KC) {
                const int32_t r = (int32_t)r0 + (q >> 3);
                kl[p] = (q & 7) << 2;
                if (r < (int32_t)Rext) rok |= 1u << p;
                rc[p] = (r < (int32_t)Rext ? r : (int32_t)Rext - 1) * ld;
                voff[p] = (uint32_t)(rc[p] + kbeg + kl[p]) * 4u;
            } else {
                const int32_t r = (int32_t)r0 + ((q % (R / 4)) << 2);
                kl[p] = q / (R / 4);
                if (r < (int32_t)Rext) rok |= 1u << p;
                rc[p] = r < (int32_t)Rext ? r : (int32_t)Rext - 4;
                voff[p] = (uint32_t)((kbeg + kl[p]) * ld + rc[p]) * 4u;
            }
        }
    }

    // Issues the loads only; the zero select of out-of-range elements happens in store_one, so no
    // instruction between here and the LDS write (a tile later) has to wait for the data.
    // Returns the validity bits of the NST float4s.
    __device__ __forceinline__ uint32_t load(int64_t t, float4 (&st)[NST]) const {
        uint32_t okm = 0;
#pragma unroll
        for (int p = 0; p < NST; ++p) {
            const int32_t k = kbeg + (int32_t)t * FX_BK + kl[p];
            if ((k < kend) && ((rok >> p) & 1u)) okm |= 1u << p;
            if constexpr (KC) {
                const int32_t kc = k < kend ? k : kend - 4;
                st[p] = *reinterpret_cast<const float4*>(P + (rc[p] + kc));
            } else {
                const int32_t kc = k < kend ? k : kend - 1;
                st[p] = *reinterpret_cast<const float4*>(P + (kc * ld + rc[p]));
            }
        }
        return okm;
    }

    // tile fully inside the matrix: uniform tile base + constant 32-bit per-lane byte offset (the
    // global_load saddr form: no per-lane address arithmetic in the loop)
    template <int p>
    __device__ __forceinline__ void load_plain(int64_t t, float4 (&st)[NST], uint32_t& okm) const {
        const int64_t tile_off = KC ? t * (FX_BK * 4) : t * (FX_BK * 4) * (int64_t)ld;
        const char* base = reinterpret_cast<const char*>(P) + tile_off;
        st[p] = *reinterpret_cast<const float4*>(base + voff[p]);
        okm = (1u << NST) - 1u;
    }

    template <int p>
    __device__ __forceinline__ void load_one(int64_t t, float4 (&st)[NST], uint32_t& okm) const {
        const int32_t k = kbeg + (int32_t)t * FX_BK + kl[p];
        if ((k < kend) && ((rok >> p) & 1u)) okm |= 1u << p;
        else okm &= ~(1u << p);
        if constexpr (KC) {
            const int32_t kc = k < kend ? k : kend - 4;
            st[p] = *reinterpret_cast<const float4*>(P + (rc[p] + kc));
        } else {
            const int32_t kc = k < kend ? k : kend - 1;
            st[p] = *reinterpret_cast<const float4*>(P + (kc * ld + rc[p]));
        }
    }

    // one LDS write instruction: component `comp` of float4 p (KC, transposing) or the whole float4
    template <int p, int comp, bool MASK>
    __device__ __forceinline__ void store_piece(float* __restrict__ T, const float4 (&st)[NST],
                                                uint32_t okm) const {
        const int q = threadIdx.x + 256 * p;
        const bool ok = MASK ? ((okm >> p) & 1u) : true;
        if constexpr (KC) {
            const int r = q >> 3, kq = (q & 7) << 2;
            const float x = comp == 0 ? st[p].x : comp == 1 ? st[p].y : comp == 2 ? st[p].z : st[p].w;
            T[(kq + comp) * LD + r] = ok ? x : 0.f;
        } else {
            const int k = q / (R / 4), r = (q % (R / 4)) << 2;
            float4 v;
            v.x = ok ? st[p].x : 0.f;
            v.y = ok ? st[p].y : 0.f;
            v.z = ok ? st[p].z : 0.f;
            v.w = ok ? st[p].w : 0.f;
            *reinterpret_cast<float4*>(T + k * LD + r) = v;
        }
    }

    template <int p>
    __device__ __forceinline__ void store_one(float* __restrict__ T, const float4 (&st)[NST],
                                              uint32_t okm) const {
        const int q = threadIdx.x + 256 * p;
        const bool ok = (okm >> p) & 1u;
        float4 v;
        v.x = ok ? st[p].x : 0.f;
        v.y = ok ? st[p].y : 0.f;
        v.z = ok ? st[p].z : 0.f;
        v.w = ok ? st[p].w : 0.f;
        if constexpr (KC) {
            const int r = q >> 3, kq = (q & 7) << 2;
            T[(kq + 0) * LD + r] = v.x;
            T[(kq + 1) * LD + r] = v.y;
            T[(kq + 2) * LD + r] = v.z;
            T[(kq + 3) * LD + r] = v.w;
        } else {
            const int k = q / (R / 4), r = (q % (R / 4)) << 2;
            *reinterpret_cast<float4*>(T + k * LD + r) = v;
        }
    }
};

template <int BM, int BN, bool A_KC, bool B_KC>
struct PipeSmem {
    static constexpr int SA = FX_BK * PipeLoader<BM, A_KC>::LD, SB = FX_BK * PipeLoader<BN, B_KC>::LD;
    static constexpr int FLOATS = 2 * SA + 2 * SB;
};

// One output tile (linear tile index L of tiles_m x tiles_n, K slab z) of the pipelined GEMM.  A
// device function so that one launch can carry tiles of more than one problem (k_gemm_f32_pair).
// TR: the MFMA is issued with its operands swapped, so the accumulators hold the TRANSPOSED 32x32
// tile — a lane owns ONE row m of C and, per group of four registers, four ADJACENT columns — and the
// epilogue reads its operands and writes C as 16-byte vectors: 4 store instructions per 32x32 tile
// instead of 16 (the drain of a launch is store-issue bound: all workgroups of a launch reach their
// epilogue together).  a*b commutes, the k order is unchanged: bit-identical results.  Needs N % 4 == 0
// and 16-byte aligned C / epilogue operands (fx_gemm_tr_ok).
template <int BM, int BN, bool A_KC, bool B_KC, bool TR = false>
__device__ __forceinline__ void fx_gemm_pipe_tile(const GemmArgs& a, const int64_t L, const int z,
                                                  float* const fx_gemm_smem) {
    FX_LAB_STAMP(0);
    using LoaderA = PipeLoader<BM, A_KC>;
    using LoaderB = PipeLoader<BN, B_KC>;
    constexpr int LDA = LoaderA::LD, LDB = LoaderB::LD;
    constexpr int NSA = LoaderA::NST, NSB = LoaderB::NST, NS = NSA + NSB;
    constexpr int MI = BM / 64, NJ = BN / 64;
    constexpr int SA = FX_BK * LDA, SB = FX_BK * LDB;
    constexpr int NG = FX_BK / 2;                      // MFMA groups (k-pairs) per tile
    float* const As0 = fx_gemm_smem;
    float* const Bs0 = fx_gemm_smem + 2 * SA;

    const int64_t nwg = (int64_t)a.tiles_m * a.tiles_n;
    int64_t T = L;
    if (nwg >= 8) {
        const int64_t q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int64_t m0 = (T / a.tiles_n) * BM;
    const int64_t n0 = (T % a.tiles_n) * BN;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int half = lane >> 5, l31 = lane & 31;
    // Fused row sums of op(A) (the bias gradient when op(A) = dZ^T): a by-product of the A fragments the
    // waves of the first tile column hold anyway — lane l sums A[row l & 31][k] over the k of its half,
    // one exact fma (x * 1 + s) per fragment beside the MFMAs, the two halves meet in one shuffle at
    // the end.  (Round 2 summed 32 LDS values per row and k-tile at the top of the tile body: the
    // n0 == 0 workgroups ran 10 % longer than the rest and ended the launch late,
    // profiles/r03_gemm_lab_b.txt.)  rs_scale = 0 for every other wave: no branch in the loop.
    const bool do_rowsum = (a.epi.rowsum != nullptr) && (n0 == 0) && (wn == 0);
    const float rs_scale = do_rowsum ? 1.f : 0.f;
    float rs[MI];
#pragma unroll
    for (int i = 0; i < MI; ++i) rs[i] = 0.f;

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int64_t nk = (kend > kbeg) ? (kend - kbeg + FX_BK - 1) / FX_BK : 0;
    if (nk > 0) {
        LoaderA la;
        LoaderB lb;
        la.init(a.A, a.lda, m0, a.M, kbeg, kend);
        lb.init(a.B, a.ldb, n0, a.N, kbeg, kend);
        float4 ra[2][NSA], rb[2][NSB];
        uint32_t oka[2], okb[2];
        oka[0] = la.load(0, ra[0]);
        okb[0] = lb.load(0, rb[0]);
        oka[1] = la.load(1, ra[1]);      // past the last tile: clamped addresses, all bits clear
        okb[1] = lb.load(1, rb[1]);
        fx_static_for<0, NSA>([&](auto p) { la.template store_one<p.value>(As0, ra[0], oka[0]); });
        fx_static_for<0, NSB>([&](auto p) { lb.template store_one<p.value>(Bs0, rb[0], okb[0]); });
        __syncthreads();
        FX_LAB_STAMP(1);
        const int foff_a = half * LDA + wm * (BM / 2) + l31;
        const int foff_b = half * LDB + wn * (BN / 2) + l31;
        float fa[2][MI], fb[2][NJ];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int i = 0; i < MI; ++i) fa[c][i] = As0[foff_a + (2 * c) * LDA + 32 * i];
#pragma unroll
            for (int j = 0; j < NJ; ++j) fb[c][j] = Bs0[foff_b + (2 * c) * LDB + 32 * j];
        }
        int s = 0;                                      // LDS stage of tile t
        // MASK = false: the tile being written to LDS (t+1) lies fully inside the matrix, its
        // registers go to LDS as they are (1 instruction per write instead of and/cmp/cndmask/write)
        auto body = [&](int64_t t, auto par, auto msk) {
            constexpr int P = decltype(par)::value;
            constexpr bool MASK = decltype(msk)::value;
            const int sn = s ^ 1;
            // (tile t+1 sits in register set P^1; past the last tile the loads are clamped)
            const uint32_t oka_n = oka[P ^ 1], okb_n = okb[P ^ 1];
            const int64_t tl = t + 2;
            const float* as = As0 + s * SA + foff_a;
            const float* bs = Bs0 + s * SB + foff_b;
            const float* asn = As0 + sn * SA + foff_a;
            const float* bsn = Bs0 + sn * SB + foff_b;
            float* wa = As0 + sn * SA;
            float* wb = Bs0 + sn * SB;

            // One k-pair group = MI*NJ MFMAs.  Its LDS work — MI+NJ fragment reads for group g+2
            // and this group's share of the refill writes — is issued ONE instruction after each
            // MFMA (measured, scripts/ubench/mfma_stream*.hip: a clump of 8 LDS instructions between
            // two groups costs the MFMA pipe ~10 %, spread out it is free with two waves per SIMD).
            fx_static_for<0, NG>([&](auto gg) {
                constexpr int g = decltype(gg)::value;
                constexpr int c = g & 1;
                constexpr int S = MI * NJ;
                constexpr int PA = A_KC ? 4 : 1, PB = B_KC ? 4 : 1;        // LDS writes per float4
                constexpr int NP = NSA * PA + NSB * PB;                     // write pieces per tile
                // groups [0, GL): the NS global loads of tile t+2 (with their address arithmetic);
                // groups [G0, G0+GW): the LDS writes of tile t+1; barrier after group NG-3
                constexpr int GL = 4, G0 = GL, GW = NG - 3 - G0;
                constexpr int llo = g < GL ? (g * NS + GL - 1) / GL : 0;
                constexpr int lhi = g < GL ? ((g + 1) * NS + GL - 1) / GL : 0;
                constexpr int lo = (g >= G0 && g < G0 + GW) ? ((g - G0) * NP + GW - 1) / GW : 0;
                constexpr int hi = (g >= G0 && g < G0 + GW) ? ((g - G0 + 1) * NP + GW - 1) / GW : 0;
                constexpr int NOPS = MI + NJ + (hi - lo) + (lhi - llo);
                float nfa[MI], nfb[NJ];
                fx_static_for<0, S>([&](auto mm) {
                    constexpr int m = decltype(mm)::value;
                    constexpr int i = m / NJ, j = m % NJ;
                    if constexpr (TR)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fb[c][j], fa[c][i], acc[i][j],
                                                                         0, 0, 0);
                    else
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j],
                                                                         0, 0, 0);
                    fx_static_for<0, NOPS>([&](auto oo) {
                        constexpr int o = decltype(oo)::value;
                        if constexpr (o % S == m) {
                            if constexpr (o < MI) {
                                if constexpr (g + 2 < NG) nfa[o] = as[(2 * g + 4) * LDA + 32 * o];
                                else nfa[o] = asn[(2 * (g + 2 - NG)) * LDA + 32 * o];
                            } else if constexpr (o < MI + NJ) {
                                constexpr int jj = o - MI;
                                if constexpr (g + 2 < NG) nfb[jj] = bs[(2 * g + 4) * LDB + 32 * jj];
                                else nfb[jj] = bsn[(2 * (g + 2 - NG)) * LDB + 32 * jj];
                            } else if constexpr (g < GL) {
                                constexpr int idx = llo + (o - MI - NJ);
                                if constexpr (MASK) {
                                    if constexpr (idx < NSA) la.template load_one<idx>(tl, ra[P], oka[P]);
                                    else lb.template load_one<idx - NSA>(tl, rb[P], okb[P]);
                                } else {
                                    if constexpr (idx < NSA) la.template load_plain<idx>(tl, ra[P], oka[P]);
                                    else lb.template load_plain<idx - NSA>(tl, rb[P], okb[P]);
                                }
                            } else {
                                constexpr int pp = lo + (o - MI - NJ);
                                if constexpr (pp < NSA * PA)
                                    la.template store_piece<pp / PA, pp % PA, MASK>(wa, ra[P ^ 1], oka_n);
                                else
                                    lb.template store_piece<(pp - NSA * PA) / PB, (pp - NSA * PA) % PB,
                                                            MASK>(wb, rb[P ^ 1], okb_n);
                            }
                        }
                    });
                    __builtin_amdgcn_sched_barrier(0);
                });
#pragma unroll
                for (int i = 0; i < MI; ++i) {
                    // (inline asm on purpose: left to the compiler the MI fmas are SLP-packed into
                    // v_pk_fma_f32, which costs the matrix pipe ~22 cycles per issue beside MFMAs —
                    // MI355X_MICROARCH.md, "price of one filler beside MFMAs")
                    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(rs[i]) : "s"(rs_scale), "v"(fa[c][i]));
                    fa[c][i] = nfa[i];
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) fb[c][j] = nfb[j];
                // Barrier once per tile, after group NG-3: every read of stage s has been issued (the
                // fragments of the last two groups were fetched in groups NG-4/NG-3) and is complete
                // (lgkmcnt(0)), every wave's writes of stage sn are complete; groups NG-2/NG-1 then
                // prefetch from sn.  Two stages are enough: nobody reads s after this barrier, and
                // the next writes into s (tile t+2's data) come after it in program order.
                if constexpr (g == NG - 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            s = sn;
        };
        // pairs in the loop, odd tail outside: a skip path inside the loop would join two different
        // "loads in flight" states at the back edge and the compiler then waits vmcnt(0) there
        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        const int64_t nk_full = (kend - kbeg) / FX_BK;           // tiles with all 32 k inside
        // plain bodies: tile t+1 (written to LDS) and tile t+2 (loaded) have all 32 k inside.  Tiles on
        // the M / N edge take them too (round 4; FX_GEMM_EDGE_PLAIN=0 restores the masked bodies): the
        // rows past the edge are loaded from clamped, in-range addresses (voff is built from rc) and
        // reach the MFMAs unmasked, but a row m >= M of A only ever feeds row m of C and a column
        // n >= N of B only column n — neither is stored (nor is its row sum).  Only the K tail has to
        // be zero.  624-wide operands (the 39 x 16 record): 10 % of the tiles of a launch were running
        // the masked bodies for their whole K loop and ended the launch late.
        const bool rows_full = (m0 + BM <= a.M) && (n0 + BN <= a.N);
        const int64_t n_plain = (rows_full || a.edge_plain) ? nk_full - 2 : 0;
        int64_t t = 0;
        for (; t + 1 < n_plain; t += 2) {
            body(t, P0{}, std::false_type{});
            body(t + 1, P1{}, std::false_type{});
        }
        for (; t + 1 < nk; t += 2) {
            body(t, P0{}, std::true_type{});
            body(t + 1, P1{}, std::true_type{});
        }
        if (t < nk) body(t, P0{}, std::true_type{});
    }
    FX_LAB_STAMP(2);

    if (a.epi.rowsum != nullptr && n0 == 0) {         // workgroup-uniform
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const float tot = rs[i] + __shfl_xor(rs[i], 32, 64);
            const int64_t m = m0 + wm * (BM / 2) + i * 32 + l31;
            if (do_rowsum && half == 0 && m < a.M) {
                if (a.split_k > 1) a.ws[(int64_t)a.split_k * a.M * a.N + (int64_t)z * a.M + m] = tot;
                else a.epi.rowsum[m] = tot;
            }
        }
    }
    if constexpr (TR) {
        // lane: row m = l31 of the wave tile; registers 4q .. 4q+3: columns 8q + 4*half + 0..3
        constexpr int NT = MI * NJ;
        const int64_t mb = m0 + wm * (BM / 2) + l31, nb = n0 + wn * (BN / 2) + 4 * half;
        if (a.split_k > 1) {
            fx_static_for<0, NT>([&](auto tt) {
                constexpr int i = decltype(tt)::value / NJ, j = decltype(tt)::value % NJ;
                const int64_t m = mb + i * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t n = nb + j * 32 + 8 * q;
                    if (m < a.M && n < a.N)
                        *reinterpret_cast<float4*>(a.ws + ((int64_t)z * a.M + m) * a.N + n) =
                            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2],
                                        acc[i][j][4 * q + 3]);
                }
            });
        } else {
            FxEpiOps4 ops[2][4];
            auto load_tile = [&](auto tt, FxEpiOps4 (&o)[4]) {
                constexpr int i = decltype(tt)::value / NJ, j = decltype(tt)::value % NJ;
                const int64_t m = mb + i * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t n = nb + j * 32 + 8 * q;
                    if (m < a.M && n < a.N) fx_epi_load4(a.epi, m, n, o[q]);
                }
            };
            load_tile(std::integral_constant<int, 0>{}, ops[0]);
            fx_static_for<0, NT>([&](auto tt) {
                constexpr int t = decltype(tt)::value;
                constexpr int i = t / NJ, j = t % NJ;
                if constexpr (t + 1 < NT) load_tile(std::integral_constant<int, t + 1>{}, ops[(t + 1) & 1]);
                const int64_t m = mb + i * 32;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t n = nb + j * 32 + 8 * q;
                    if (m < a.M && n < a.N) {
                        const float4 v = make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1],
                                                     acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
                        *reinterpret_cast<float4*>(a.C + m * a.ldc + n) =
                            fx_epi_apply4(a.epi, v, m, n, ops[t & 1][q]);
                    }
                }
            });
        }
    } else {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int64_t n = n0 + wn * (BN / 2) + j * 32 + l31;
                if (n >= a.N) continue;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t m = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m >= a.M) continue;
                    if (a.split_k > 1) {
                        a.ws[((int64_t)z * a.M + m) * a.N + n] = acc[i][j][r];
                    } else {
                        a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc[i][j][r], m, n);
                    }
                }
            }
        }
    }
#ifdef FX_GEMM_LAB
    if (a.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        FX_LAB_STAMP(3);
        if (threadIdx.x == 0) {
            const int64_t w = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
            a.trace[w + 4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_REG_HW_ID
            a.trace[w + 5] = __builtin_amdgcn_s_getreg((31 << 11) | 20);    // HW_REG_XCC_ID
        }
    }
#endif
}

template <int BM, int BN, bool A_KC, bool B_KC, int W = 2, bool TR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
void k_gemm_f32_pipe(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[PipeSmem<BM, BN, A_KC, B_KC>::FLOATS];
    fx_gemm_pipe_tile<BM, BN, A_KC, B_KC, TR>(a, blockIdx.x, blockIdx.y, smem);
}

// Two independent GEMMs in ONE launch (fx_gemm_f32_batch): the weight gradient dW = dZ^T X (problem 1,
// operands m-/n-contiguous, split-K slabs) and the input gradient dX = dZ W (problem 2) of a layer
// share dZ and neither depends on the other.  Launched separately each pays its own ramp — all
// workgroups resident at once, prologue loads and epilogue stores in lock step, ~6 us of idle matrix
// pipes per launch (K sweep in profiles/r02_gemm_probe.txt); in one grid the second problem's
// workgroups start as the first one's retire, and the 624-wide CrossNet shapes (640 tiles on 1024
// slots) no longer leave a third of the CUs one workgroup short.
template <int BM, int BN, bool A1, bool B1, bool A2, bool B2, int W, bool TR = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
void k_gemm_f32_pair(GemmArgs a1, GemmArgs a2) {
    constexpr int F1 = PipeSmem<BM, BN, A1, B1>::FLOATS, F2 = PipeSmem<BM, BN, A2, B2>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[F1 > F2 ? F1 : F2];
    const int64_t n1 = (int64_t)a1.tiles_m * a1.tiles_n, w1 = n1 * a1.split_k;
    const int64_t L = blockIdx.x;
    if (L < w1) {
        fx_gemm_pipe_tile<BM, BN, A1, B1, TR>(a1, L % n1, (int)(L / n1), smem);
    } else {
        const int64_t n2 = (int64_t)a2.tiles_m * a2.tiles_n, L2 = L - w1;
        fx_gemm_pipe_tile<BM, BN, A2, B2, TR>(a2, L2 % n2, (int)(L2 / n2), smem);
    }
}

// Up to FX_MULTI_MAX independent GEMMs in ONE launch on 128-row tiles, two workgroups per CU
// (fx_gemm_f32_batch).  Round 3 timelines (profiles/r03_gemm_lab_a.txt): a 128x128 workgroup — one wave per
// SIMD with four accumulators — streams its K loop at 0.91-0.95 of the matrix-pipe peak on its own, while
// the four 64x64 workgroups of a CU (one accumulator per wave) finish between 50 and 81 us of an 81-us
// launch: the SIMD arbitrates oldest-first, the early finishers leave the late ones alone on the pipe at
// a third of its rate.  So: big tiles, and a SECOND problem's workgroup as the co-resident instead of
// three more of the same — the dW and dX products of a layer (and, for DCNv2's parallel structure, the
// cross and the deep layer of the same depth) fill each other's prologue / epilogue gaps.
// cfg bit 0: A k-contiguous, bit 1: B k-contiguous, bit 2: 128x64 tile (else 128x128).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_f32_multi(MultiArgs a) {
    constexpr int F0 = PipeSmem<128, 128, false, false>::FLOATS, F1 = PipeSmem<128, 128, true, true>::FLOATS,
                  F2 = PipeSmem<128, 128, true, false>::FLOATS;
    constexpr int FM = F0 > F1 ? (F0 > F2 ? F0 : F2) : (F1 > F2 ? F1 : F2);
    __shared__ __attribute__((aligned(16))) float smem[FM];
    int i = 0;
    while (i + 1 < a.n && (int32_t)blockIdx.x >= a.start[i + 1]) ++i;
    // the problem's arguments are read through the kernarg segment pointer (uniform scalar loads):
    // indexing the by-value struct with a run-time index made the compiler copy it to scratch
    const MultiArgs* ka = (const MultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const GemmArgs& g = ka->p[i];
    int64_t L = (int64_t)blockIdx.x - a.start[i];
    const int64_t nt = (int64_t)g.tiles_m * g.tiles_n;
    const int z = (int)(L / nt);
    L -= (int64_t)z * nt;
    switch (ka->cfg[i]) {
        case 0: fx_gemm_pipe_tile<128, 128, false, false, true>(g, L, z, smem); break;
        case 1: fx_gemm_pipe_tile<128, 128, true, false, true>(g, L, z, smem); break;
        case 2: fx_gemm_pipe_tile<128, 128, false, true, true>(g, L, z, smem); break;
        case 3: fx_gemm_pipe_tile<128, 128, true, true, true>(g, L, z, smem); break;
        case 4: fx_gemm_pipe_tile<128, 64, false, false, true>(g, L, z, smem); break;
        case 5: fx_gemm_pipe_tile<128, 64, true, false, true>(g, L, z, smem); break;
        case 6: fx_gemm_pipe_tile<128, 64, false, true, true>(g, L, z, smem); break;
        default: fx_gemm_pipe_tile<128, 64, true, true, true>(g, L, z, smem); break;
    }
}

// (Round 2 experiment, removed again: a variant that kept k-contiguous operands in their global
// layout in LDS — T[r][36], one ds_write_b128 per staging float4, one ds_read_b128 per lane and 8-k
// block, i.e. 12 instead of 48 LDS instructions per 16 MFMAs of a 64x64 tile — measured the SAME
// as this kernel on every tower shape (78.9 vs 78.7 us at 4096x1024x1024, identical K slope,
// profiles/r02_gemm_probe.txt): the 64x64 loop is not bound by LDS traffic or instruction issue.)
static int fx_gemm_pipe_mode() {   // FX_GEMM_PIPE=0 falls back to the unpipelined kernel (A/B runs)
    static const int mode = []() {
        const char* e = getenv("FX_GEMM_PIPE");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

static int fx_gemm_tr_mode() {     // FX_GEMM_TR=0: 4-byte epilogue stores everywhere (A/B runs)
    static const int mode = []() {
        const char* e = getenv("FX_GEMM_TR");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

static bool fx_al16(const void* p, int64_t ld) {
    return p == nullptr || ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (ld & 3) == 0);
}

// the 16-byte epilogue (TR) applies: every vector access of fx_epilogue4 / the slab stores is aligned
static bool fx_gemm_tr_ok(const GemmArgs& a) {
    const fx_gemm_epilogue& e = a.epi;
    return fx_gemm_tr_mode() && (a.N & 3) == 0 && fx_al16(a.C, a.ldc) && fx_al16(e.bias, 0) &&
           fx_al16(e.zout, e.ldz) && fx_al16(e.mul, e.ldmul) && fx_al16(e.mask, e.ldmask) &&
           fx_al16(e.add, e.ldadd) && (a.split_k == 1 || fx_al16(a.ws, 0));
}

template <int BM, int BN, bool A_KC, bool B_KC, bool TR>
static int fx_gemm_launch_pipe_tr(dim3 grid, hipStream_t s, const GemmArgs& a) {
    if constexpr (BM * BN <= 64 * 64) {
        // 64x64 tiles need ~110 VGPRs: 4 waves/SIMD = 4 workgroups per CU (LDS 4 x 34 KB), so the
        // 1024 tiles of a 4096 x 1024 layer are all resident in ONE round (2 per CU took two; the
        // 2- and 3-wave builds of round 2, FX_GEMM_W64, measured slower and are gone)
        hipLaunchKernelGGL((k_gemm_f32_pipe<BM, BN, A_KC, B_KC, 4, TR>), grid, dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((k_gemm_f32_pipe<BM, BN, A_KC, B_KC, 2, TR>), grid, dim3(256), 0, s, a);
    }
    return FX_OK;
}

template <int BM, int BN, bool A_KC, bool B_KC>
static int fx_gemm_launch_pipe(dim3 grid, hipStream_t s, const GemmArgs& a) {
    if (fx_gemm_tr_ok(a)) return fx_gemm_launch_pipe_tr<BM, BN, A_KC, B_KC, true>(grid, s, a);
    return fx_gemm_launch_pipe_tr<BM, BN, A_KC, B_KC, false>(grid, s, a);
}

template <int BM, int BN>
static int fx_gemm_dispatch_pipe(bool a_kc, bool b_kc, dim3 grid, hipStream_t s, const GemmArgs& a) {
    if (a_kc && b_kc) return fx_gemm_launch_pipe<BM, BN, true, true>(grid, s, a);
    if (a_kc) return fx_gemm_launch_pipe<BM, BN, true, false>(grid, s, a);
    if (b_kc) return fx_gemm_launch_pipe<BM, BN, false, true>(grid, s, a);
    return fx_gemm_launch_pipe<BM, BN, false, false>(grid, s, a);
}

__global__ __launch_bounds__(256) void k_splitk_reduce(GemmArgs a) {
    const int64_t total = a.M * a.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        float s = 0.f;
        for (int z = 0; z < a.split_k; ++z) s += a.ws[(int64_t)z * total + i];
        const int64_t m = i / a.N, n = i - m * a.N;
        a.C[m * a.ldc + n] = fx_epilogue(a.epi, s, m, n);
    }
    if (a.epi.rowsum) {
        const float* rs = a.ws + (int64_t)a.split_k * total;
        for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < a.M;
             m += (int64_t)gridDim.x * 256) {
            float s = 0.f;
            for (int z = 0; z < a.split_k; ++z) s += rs[(int64_t)z * a.M + m];
            a.epi.rowsum[m] = s;
        }
    }
}


// The same sums on 16-byte vectors (N % 4 == 0, everything 16-byte aligned: fx_gemm_tr_ok): the slab loads
// of a vector are issued together (SK <= 8 of them: the split rules' range) and added in slab order, so the
// result is bit for bit k_splitk_reduce's.  4096 x 1024 x 1024's weight gradient (8 slabs of 4 MB): 7.4 us
// -> round 4's A/B in profiles/.
template <int SK>
__device__ __forceinline__ void fx_splitk_reduce_v4_body(const GemmArgs& a, int64_t bx, int64_t gx) {
    const int64_t total = a.M * a.N, nv = total >> 2, n4 = a.N >> 2;
    const int sk = SK > 0 ? SK : a.split_k;
    for (int64_t i = bx * 256 + threadIdx.x; i < nv; i += gx * 256) {
        const float4* w = reinterpret_cast<const float4*>(a.ws) + i;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (SK > 0) {
            float4 v[SK];
#pragma unroll
            for (int z = 0; z < SK; ++z) v[z] = w[(int64_t)z * nv];
#pragma unroll
            for (int z = 0; z < SK; ++z) { s.x += v[z].x; s.y += v[z].y; s.z += v[z].z; s.w += v[z].w; }
        } else {
            int z = 0;
            for (; z + 4 <= sk; z += 4) {          // four independent loads at a time, added in slab order
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = w[(int64_t)(z + u) * nv];
#pragma unroll
                for (int u = 0; u < 4; ++u) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
            for (; z < sk; ++z) {
                const float4 v = w[(int64_t)z * nv];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        const int64_t m = i / n4, n = (i - m * n4) << 2;
        FxEpiOps4 o;
        fx_epi_load4(a.epi, m, n, o);
        *reinterpret_cast<float4*>(a.C + m * a.ldc + n) = fx_epi_apply4(a.epi, s, m, n, o);
    }
    if (a.epi.rowsum) {
        const float* rs = a.ws + (int64_t)sk * total;
        for (int64_t m = bx * 256 + threadIdx.x; m < a.M;
             m += gx * 256) {
            float r = 0.f;
            int z = 0;
            for (; z + 8 <= sk; z += 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = rs[(int64_t)(z + u) * a.M + m];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += v[u];
            }
            for (; z < sk; ++z) r += rs[(int64_t)z * a.M + m];
            a.epi.rowsum[m] = r;
        }
    }
}

template <int SK>
__global__ __launch_bounds__(256) void k_splitk_reduce_v4(GemmArgs a) {
    fx_splitk_reduce_v4_body<SK>(a, (int64_t)blockIdx.x, (int64_t)gridDim.x);
}

// (round 6) the slab reduces of every weight gradient of ONE multi-problem GEMM launch in one launch (DCNv2's
// cross + deep pairs: 7 reduce launches per step -> 4): workgroups [start[i], start[i + 1]) take problem i and
// run the single-problem body on it — same sums, same order, same bits.
struct ReduceMultiArgs {
    GemmArgs p[FX_MULTI_MAX];
    int32_t start[FX_MULTI_MAX + 1];
    int32_t n;
};
__global__ __launch_bounds__(256) void k_splitk_reduce_v4_multi(ReduceMultiArgs ma) {
    int i = 0;
#pragma unroll
    for (int q = 1; q < FX_MULTI_MAX; ++q)
        if (q < ma.n && (int)blockIdx.x >= ma.start[q]) i = q;
    const int64_t bx = (int64_t)blockIdx.x - ma.start[i], gx = (int64_t)ma.start[i + 1] - ma.start[i];
    // (indexing p[] by a runtime value would copy the 200-byte argument block to scratch: select by branches)
#define FX_RM_CASE(Q)                                                                        \
    if (i == Q) {                                                                            \
        const GemmArgs& a = ma.p[Q];                                                         \
        switch (a.split_k) {                                                                 \
            case 2: fx_splitk_reduce_v4_body<2>(a, bx, gx); break;                           \
            case 4: fx_splitk_reduce_v4_body<4>(a, bx, gx); break;                           \
            case 8: fx_splitk_reduce_v4_body<8>(a, bx, gx); break;                           \
            default: fx_splitk_reduce_v4_body<0>(a, bx, gx); break;                          \
        }                                                                                    \
        return;                                                                              \
    }
    FX_RM_CASE(0)
    FX_RM_CASE(1)
    FX_RM_CASE(2)
    FX_RM_CASE(3)
#undef FX_RM_CASE
}

// many slabs over a small output (skinny weight gradients with K = B*L): EL elements x 256/EL slab
// lanes per workgroup, fixed LDS tree over the slab lanes (deterministic)
template <int EL>
__global__ __launch_bounds__(256) void k_splitk_reduce_wide(GemmArgs a) {
    constexpr int ZL = 256 / EL;
    __shared__ float red[256];
    const int ii = threadIdx.x % EL, zi = threadIdx.x / EL;
    const int64_t total = a.M * a.N;
    const int64_t i = (int64_t)blockIdx.x * EL + ii;
    float s = 0.f;
    if (i < total) {
        int z = zi;
        for (; z + 7 * ZL < a.split_k; z += 8 * ZL) {      // 8 independent loads in flight,
            float v[8];                                     // summed in slab order
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = a.ws[(int64_t)(z + u * ZL) * total + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < a.split_k; z += ZL) s += a.ws[(int64_t)z * total + i];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = ZL >> 1; h > 0; h >>= 1) {
        if (zi < h) red[threadIdx.x] += red[threadIdx.x + h * EL];
        __syncthreads();
    }
    if (zi == 0 && i < total) {
        const int64_t m = i / a.N, n = i - m * a.N;
        a.C[m * a.ldc + n] = fx_epilogue(a.epi, red[ii], m, n);
    }
    if (a.epi.rowsum && blockIdx.x == 0) {     // block-uniform
        // fused bias gradient: rowsum[m] = sum over the slabs' row sums.  With hundreds of slabs a
        // one-thread-per-row loop is a chain of dependent loads (64 us for 256 slabs): all 256
        // threads work, Mp (= pow2 >= M, M <= 256) rows x 256/Mp slab lanes, same fixed LDS tree
        const float* rs = a.ws + (int64_t)a.split_k * total;
        int mp_log2 = 0;
        while ((1 << mp_log2) < a.M) ++mp_log2;
        const int Mp = 1 << mp_log2, ZR = 256 >> mp_log2;
        const int m = threadIdx.x & (Mp - 1), zr = threadIdx.x >> mp_log2;
        __syncthreads();                         // red[] is reused: the output tree is fully read
        float r = 0.f;
        if (m < a.M) {
            int z = zr;
            for (; z + 7 * ZR < a.split_k; z += 8 * ZR) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = rs[(int64_t)(z + u * ZR) * a.M + m];
#pragma unroll
                for (int u = 0; u < 8; ++u) r += v[u];
            }
            for (; z < a.split_k; z += ZR) r += rs[(int64_t)z * a.M + m];
        }
        red[threadIdx.x] = r;
        __syncthreads();
        for (int h = ZR >> 1; h > 0; h >>= 1) {
            if (zr < h) red[threadIdx.x] += red[threadIdx.x + h * Mp];
            __syncthreads();
        }
        if (zr == 0 && m < a.M) a.epi.rowsum[m] = red[m];
    }
}

static int fx_splitk_v4_mode() {     // FX_SPLITK_V4=0: the 4-byte slab reduce (A/B runs)
    static const int mode = []() {
        const char* e = getenv("FX_SPLITK_V4");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

static void fx_launch_splitk_reduce(const GemmArgs& a, hipStream_t s) {
    const int64_t total = a.M * a.N;
    if (a.split_k >= 32 && total <= 65536 && a.M <= 256) {
        if (a.split_k >= 128 && total <= 2048)   // few outputs, very many slabs: more slab lanes
            hipLaunchKernelGGL(k_splitk_reduce_wide<8>, dim3((unsigned)fx_ceil_div(total, 8)),
                               dim3(256), 0, s, a);
        else
            hipLaunchKernelGGL(k_splitk_reduce_wide<32>, dim3((unsigned)fx_ceil_div(total, 32)),
                               dim3(256), 0, s, a);
    } else if (fx_splitk_v4_mode() && fx_gemm_tr_ok(a) && (total & 3) == 0) {
        int64_t blocks = fx_ceil_div(total >> 2, 256);      // one vector per thread up to 4096 workgroups
        if (blocks > 4096) blocks = 4096;
        const dim3 g((unsigned)blocks), b(256);
        switch (a.split_k) {
            case 2: hipLaunchKernelGGL(k_splitk_reduce_v4<2>, g, b, 0, s, a); break;
            case 4: hipLaunchKernelGGL(k_splitk_reduce_v4<4>, g, b, 0, s, a); break;
            case 8: hipLaunchKernelGGL(k_splitk_reduce_v4<8>, g, b, 0, s, a); break;
            case 16: hipLaunchKernelGGL(k_splitk_reduce_v4<16>, g, b, 0, s, a); break;   // (the DIN tower)
            default: hipLaunchKernelGGL(k_splitk_reduce_v4<0>, g, b, 0, s, a); break;
        }
    } else {
        int64_t blocks = fx_ceil_div(total, 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
}

static bool fx_splitk_reduce_is_v4(const GemmArgs& a) {
    const int64_t total = a.M * a.N;
    return !(a.split_k >= 32 && total <= 65536 && a.M <= 256) && fx_splitk_v4_mode() && fx_gemm_tr_ok(a) &&
           (total & 3) == 0;
}

// every split-K problem of a multi-problem launch: one reduce launch when two or more of them take the vector
// kernel (FX_REDUCE_MULTI=0: one launch each, as in rounds 3 - 5)
static void fx_launch_splitk_reduces(const GemmArgs* p, int n, hipStream_t s) {
    static const bool multi = []() {
        const char* e = getenv("FX_REDUCE_MULTI");
        return !(e && atoi(e) == 0);
    }();
    ReduceMultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    int cnt = 0;
    int64_t wgs = 0;
    for (int i = 0; i < n && multi; ++i)
        if (p[i].split_k > 1 && fx_splitk_reduce_is_v4(p[i])) {
            int64_t blocks = fx_ceil_div((p[i].M * p[i].N) >> 2, 256);
            if (blocks > 4096) blocks = 4096;
            ma.p[cnt] = p[i];
            ma.start[cnt] = (int32_t)wgs;
            wgs += blocks;
            ++cnt;
        }
    if (cnt >= 2) {
        for (int q = cnt; q <= FX_MULTI_MAX; ++q) ma.start[q] = (int32_t)wgs;
        ma.n = cnt;
        hipLaunchKernelGGL(k_splitk_reduce_v4_multi, dim3((unsigned)wgs), dim3(256), 0, s, ma);
    }
    for (int i = 0; i < n; ++i)
        if (p[i].split_k > 1 && !(cnt >= 2 && fx_splitk_reduce_is_v4(p[i]))) fx_launch_splitk_reduce(p[i], s);
}

// ---------------------------------------------------------------------------------------------
// Skinny shapes.  Every CTR tower ends in Linear(hidden -> 1): its forward (N = 1), weight
// gradient (M = 1) and input gradient (K = 1) would each occupy a full 128-wide MFMA tile per
// block for one useful row/column, so they run as bandwidth-bound kernels instead.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fx_a_at(const GemmArgs& a, int ta, int64_t m, int64_t k) {
    return ta ? a.A[k * a.lda + m] : a.A[m * a.lda + k];
}
__device__ __forceinline__ float fx_b_at(const GemmArgs& a, int tb, int64_t k, int64_t n) {
    return tb ? a.B[n * a.ldb + k] : a.B[k * a.ldb + n];
}

// K <= 8: one thread per output element
__global__ __launch_bounds__(256) void k_gemm_small_k(GemmArgs a, int ta, int tb) {
    const int64_t total = a.M * a.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / a.N, n = i - m * a.N;
        float acc = 0.f;
        for (int64_t k = 0; k < a.K; ++k) acc = fmaf(fx_a_at(a, ta, m, k), fx_b_at(a, tb, k, n), acc);
        a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc, m, n);
    }
}

// K <= 8 with 4 | N (the input gradient of a Linear(hidden -> 1) head: an outer product that writes
// M x N floats and reads the same amount of ReLU mask): one thread per 4 output columns, 16-byte stores,
// no 64-bit division per element.  Pure HBM stream.
__global__ __launch_bounds__(256) void k_gemm_small_k_v4(GemmArgs a, int ta, int tb) {
    const uint32_t n4 = (uint32_t)(a.N >> 2);
    const uint32_t total = (uint32_t)(a.M * n4);              // < 2^31 (checked by the launcher)
    const fx_gemm_epilogue& e = a.epi;
    // the tower case: nothing but the ReLU mask of the layer below -> one 16-byte mask load
    const bool mask_only = e.mask && !e.bias && !e.zout && e.act == 0 && !e.mul && !e.add &&
                           (e.ldmask & 3) == 0 && (reinterpret_cast<uintptr_t>(e.mask) & 15) == 0;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const uint32_t mi = i / n4;
        const int64_t m = mi, n = (int64_t)(i - mi * n4) << 2;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t k = 0; k < a.K; ++k) {
            const float x = fx_a_at(a, ta, m, k);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = fmaf(x, fx_b_at(a, tb, k, n + c), acc[c]);
        }
        float4 o;
        if (mask_only) {
            const float4 mk = *reinterpret_cast<const float4*>(e.mask + m * e.ldmask + n);
            o.x = mk.x > 0.f ? acc[0] : 0.f;
            o.y = mk.y > 0.f ? acc[1] : 0.f;
            o.z = mk.z > 0.f ? acc[2] : 0.f;
            o.w = mk.w > 0.f ? acc[3] : 0.f;
        } else {
            o.x = fx_epilogue(e, acc[0], m, n);
            o.y = fx_epilogue(e, acc[1], m, n + 1);
            o.z = fx_epilogue(e, acc[2], m, n + 2);
            o.w = fx_epilogue(e, acc[3], m, n + 3);
        }
        *reinterpret_cast<float4*>(a.C + m * a.ldc + n) = o;
    }
}

// N <= 4, A [M,K] and B [N,K] k-contiguous, 4 | K, any K: one wave per output row, a lane takes a float4
// every 256 floats, four loads in flight per lane (the scalar kernel below issues one dependent 4-byte
// load per iteration: 1.7 TB/s on the 4096 x 1024 head of the towers)
__global__ __launch_bounds__(256) void k_gemm_small_n_wide(GemmArgs a) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < a.M; m += waves) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* arow = a.A + m * a.lda;
        int64_t k = (int64_t)lane * 4;
        for (; k + 768 < a.K; k += 1024) {
            float4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) x[u] = *reinterpret_cast<const float4*>(arow + k + 256 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    if (n < a.N) {
                        const float4 w = *reinterpret_cast<const float4*>(a.B + n * a.ldb + k + 256 * u);
                        acc[n] = fmaf(x[u].x, w.x, acc[n]);
                        acc[n] = fmaf(x[u].y, w.y, acc[n]);
                        acc[n] = fmaf(x[u].z, w.z, acc[n]);
                        acc[n] = fmaf(x[u].w, w.w, acc[n]);
                    }
        }
        for (; k < a.K; k += 256) {
            const float4 x = *reinterpret_cast<const float4*>(arow + k);
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < a.N) {
                    const float4 w = *reinterpret_cast<const float4*>(a.B + n * a.ldb + k);
                    acc[n] = fmaf(x.x, w.x, acc[n]);
                    acc[n] = fmaf(x.y, w.y, acc[n]);
                    acc[n] = fmaf(x.z, w.z, acc[n]);
                    acc[n] = fmaf(x.w, w.w, acc[n]);
                }
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = fx_wave_sum(acc[n]);
        if (lane == 0)
            for (int n = 0; n < a.N; ++n) a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc[n], m, n);
    }
}

// N <= 4, A stored [M,K]: one wave per output row, lanes stride k (coalesced), xor reduction
__global__ __launch_bounds__(256) void k_gemm_small_n(GemmArgs a, int tb) {
    const int lane = threadIdx.x & 63;
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); m < a.M; m += waves) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* arow = a.A + m * a.lda;
        for (int64_t k = lane; k < a.K; k += 64) {
            const float x = arow[k];
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n < a.N) acc[n] = fmaf(x, fx_b_at(a, tb, k, n), acc[n]);
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[n] = fx_wave_sum(acc[n]);
        if (lane == 0)
            for (int n = 0; n < a.N; ++n) a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc[n], m, n);
    }
}

// M <= 4, A stored [K,M], B stored [K,N]: Np = min(256, pow2 >= N) column lanes x 256/Np row lanes
// per workgroup, K split over blockIdx.y into workspace slabs (reduced, with the epilogue, by
// k_splitk_reduce).  Narrow outputs (the 64 -> 1 head of the DIN attention MLP has N = 64 and
// K = B*L = 204800) keep all 256 lanes busy through the row lanes.
// fused bias gradient of the skinny weight-gradient kernels (M <= 4): slab z's sum of column m of
// A over [kbeg, kend).  All 256 threads of the block take part (a single thread per row made the
// k_chunk loads one dependent chain: 40 us for a 400-deep chunk); fixed LDS tree.
__device__ __forceinline__ void fx_small_m_rowsum(const GemmArgs& a, int z, int64_t kbeg,
                                                  int64_t kend) {
    __shared__ float rs[256];
    for (int m = 0; m < (int)a.M; ++m) {            // block-uniform (M <= 4)
        float r = 0.f;
        for (int64_t k = kbeg + threadIdx.x; k < kend; k += 256) r += a.A[k * a.lda + m];
        rs[threadIdx.x] = r;
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) rs[threadIdx.x] += rs[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0)
            a.ws[(int64_t)a.split_k * a.M * a.N + (int64_t)z * a.M + m] = rs[0];
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_gemm_small_m(GemmArgs a, int np_log2) {
    __shared__ float red[4][256];
    const int Np = 1 << np_log2;
    const int tx = threadIdx.x & (Np - 1), ty = threadIdx.x >> np_log2;
    const int lanes = 256 >> np_log2;
    const int64_t n = (int64_t)blockIdx.x * Np + tx;
    const int z = blockIdx.y;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
    if (a.epi.rowsum && blockIdx.x == 0) fx_small_m_rowsum(a, z, kbeg, kend);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (n < a.N) {
        for (int64_t k = kbeg + ty; k < kend; k += lanes) {
            const float b = a.B[k * a.ldb + n];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < a.M) acc[m] = fmaf(a.A[k * a.lda + m], b, acc[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) red[m][threadIdx.x] = acc[m];
    __syncthreads();
    for (int s2 = lanes >> 1; s2 > 0; s2 >>= 1) {
        if (ty < s2) {
#pragma unroll
            for (int m = 0; m < 4; ++m) red[m][threadIdx.x] += red[m][threadIdx.x + (s2 << np_log2)];
        }
        __syncthreads();
    }
    if (ty == 0 && n < a.N)
        for (int m = 0; m < a.M; ++m) a.ws[((int64_t)z * a.M + m) * a.N + n] = red[m][tx];
}

// vectorised M <= 4 variant (N % 4 == 0, 16-B aligned B rows): CG float4 column groups x 256/CG
// row lanes per workgroup (CG = 64 for N >= 256, fewer for narrow outputs such as the 64-wide DIN
// attention layer, K = B*L = 204800), two rows in flight per lane, LDS combine of the row lanes
template <int CG_LOG2>
__global__ __launch_bounds__(256) void k_gemm_small_m_v4(GemmArgs a) {
    constexpr int CG = 1 << CG_LOG2, RL = 256 / CG;
    __shared__ float4 red[4][256];
    const int tx = threadIdx.x & (CG - 1), ty = threadIdx.x >> CG_LOG2;
    const int64_t n = ((int64_t)blockIdx.x * CG + tx) * 4;
    const int z = blockIdx.y;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
    if (a.epi.rowsum && blockIdx.x == 0) fx_small_m_rowsum(a, z, kbeg, kend);
    float4 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = make_float4(0.f, 0.f, 0.f, 0.f);
    auto fma_row = [&](const float4& b, int64_t k) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (m < a.M) {
                const float w = a.A[k * a.lda + m];
                acc[m].x = fmaf(w, b.x, acc[m].x);
                acc[m].y = fmaf(w, b.y, acc[m].y);
                acc[m].z = fmaf(w, b.z, acc[m].z);
                acc[m].w = fmaf(w, b.w, acc[m].w);
            }
        }
    };
    if (n < a.N) {
        int64_t k = kbeg + ty;
        for (; k + RL < kend; k += 2 * RL) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.B + k * a.ldb + n);
            const float4 b1 = *reinterpret_cast<const float4*>(a.B + (k + RL) * a.ldb + n);
            fma_row(b0, k);
            fma_row(b1, k + RL);
        }
        for (; k < kend; k += RL) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.B + k * a.ldb + n);
            fma_row(b0, k);
        }
    }
#pragma unroll
    for (int m = 0; m < 4; ++m) red[m][threadIdx.x] = acc[m];
    __syncthreads();
    if (ty == 0 && n < a.N) {
        for (int m = 0; m < a.M; ++m) {
            float4 r = red[m][tx];
            for (int y = 1; y < RL; ++y) {          // fixed order over the row lanes
                const float4 q = red[m][tx + y * CG];
                r.x += q.x;
                r.y += q.y;
                r.z += q.z;
                r.w += q.w;
            }
            *reinterpret_cast<float4*>(a.ws + ((int64_t)z * a.M + m) * a.N + n) = r;
        }
    }
}

// Backward of a Linear(hidden -> 1) head in ONE pass over the hidden activations: the weight gradient
// dW[1, N] = sum_k dz[k] x[k, :] (k_gemm_small_m_v4 with M = 1: same loop, same slab order -> same bits)
// and the input gradient dX[k, :] = dz[k] W[:] with the ReLU mask x[k, :] > 0 — x IS the mask, so the
// row that was just loaded for dW also decides and the product leaves as one 16-byte store.  Two
// launches (k_gemm_small_m_v4 + k_gemm_small_k_v4: x streamed twice) become one.
struct HeadBwdArgs {
    GemmArgs dw;          // A = dz [K, 1] (lda), B = x [K, N] (ldb), ws slabs, rowsum
    float* dx;            // [K, N] (ldx)
    int64_t ldx;
    const float* w;       // [N]
    int32_t use_mask;
};

template <int CG_LOG2>
__global__ __launch_bounds__(256) void k_head_bwd_v4(HeadBwdArgs h) {
    const GemmArgs& a = h.dw;
    constexpr int CG = 1 << CG_LOG2, RL = 256 / CG;
    __shared__ float4 red[256];
    const int tx = threadIdx.x & (CG - 1), ty = threadIdx.x >> CG_LOG2;
    const int64_t n = ((int64_t)blockIdx.x * CG + tx) * 4;
    const int z = blockIdx.y;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
    if (a.epi.rowsum && blockIdx.x == 0) fx_small_m_rowsum(a, z, kbeg, kend);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < a.N) {
        const float4 wv = *reinterpret_cast<const float4*>(h.w + n);
        const bool um = h.use_mask != 0;
        auto row = [&](const float4& b, int64_t k) {
            const float d = a.A[k * a.lda];
            acc.x = fmaf(d, b.x, acc.x);
            acc.y = fmaf(d, b.y, acc.y);
            acc.z = fmaf(d, b.z, acc.z);
            acc.w = fmaf(d, b.w, acc.w);
            float4 o = make_float4(d * wv.x, d * wv.y, d * wv.z, d * wv.w);
            if (um) {
                o.x = b.x > 0.f ? o.x : 0.f;
                o.y = b.y > 0.f ? o.y : 0.f;
                o.z = b.z > 0.f ? o.z : 0.f;
                o.w = b.w > 0.f ? o.w : 0.f;
            }
            *reinterpret_cast<float4*>(h.dx + k * h.ldx + n) = o;
        };
        int64_t k = kbeg + ty;
        for (; k + RL < kend; k += 2 * RL) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.B + k * a.ldb + n);
            const float4 b1 = *reinterpret_cast<const float4*>(a.B + (k + RL) * a.ldb + n);
            row(b0, k);
            row(b1, k + RL);
        }
        for (; k < kend; k += RL) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.B + k * a.ldb + n);
            row(b0, k);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && n < a.N) {
        float4 r = red[tx];
        for (int y = 1; y < RL; ++y) {          // fixed order over the row lanes
            const float4 q = red[tx + y * CG];
            r.x += q.x;
            r.y += q.y;
            r.z += q.z;
            r.w += q.w;
        }
        *reinterpret_cast<float4*>(a.ws + (int64_t)z * a.N + n) = r;
    }
}

// N <= 4, A stored [M,K] with K <= 256, K % 4 == 0, 16-B aligned rows (the 64 -> 1 attention output
// layer over B*L rows): K/4 lanes read one row as float4s, 64/(K/4) rows per wave instruction
template <int LPR_LOG2>
__global__ __launch_bounds__(256) void k_gemm_small_n_v4(GemmArgs a, int tb) {
    constexpr int LPR = 1 << LPR_LOG2, RPW = 64 / LPR;     // lanes per row, rows per wave
    const int lane = threadIdx.x & 63;
    const int sub = lane & (LPR - 1), rw = lane >> LPR_LOG2;
    const int kq = sub * 4;
    float4 w[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        w[n] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < a.N && kq < a.K) {
            w[n].x = fx_b_at(a, tb, kq + 0, n);
            w[n].y = fx_b_at(a, tb, kq + 1, n);
            w[n].z = fx_b_at(a, tb, kq + 2, n);
            w[n].w = fx_b_at(a, tb, kq + 3, n);
        }
    }
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t waves = (int64_t)gridDim.x * 4;
    for (int64_t m0 = wave * RPW; m0 < a.M; m0 += waves * RPW) {
        const int64_t m = m0 + rw;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (m < a.M && kq < a.K) x = *reinterpret_cast<const float4*>(a.A + m * a.lda + kq);
        float acc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            acc[n] = fmaf(x.w, w[n].w, fmaf(x.z, w[n].z, fmaf(x.y, w[n].y, x.x * w[n].x)));
#pragma unroll
            for (int off = LPR >> 1; off > 0; off >>= 1) acc[n] += __shfl_xor(acc[n], off, 64);
        }
        if (sub == 0 && m < a.M)
            for (int n = 0; n < a.N; ++n) a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc[n], m, n);
    }
}

template <int BM, int BN, bool A_KC, bool B_KC>
static void fx_gemm_dispatch_vec(bool av, bool bv, dim3 grid, hipStream_t s, const GemmArgs& a) {
    if (av && bv)
        hipLaunchKernelGGL((k_gemm_f32<BM, BN, A_KC, B_KC, true, true>), grid, dim3(256), 0, s, a);
    else if (av)
        hipLaunchKernelGGL((k_gemm_f32<BM, BN, A_KC, B_KC, true, false>), grid, dim3(256), 0, s, a);
    else if (bv)
        hipLaunchKernelGGL((k_gemm_f32<BM, BN, A_KC, B_KC, false, true>), grid, dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((k_gemm_f32<BM, BN, A_KC, B_KC, false, false>), grid, dim3(256), 0, s, a);
}

template <int BM, int BN>
static void fx_gemm_dispatch_layout(bool a_kc, bool b_kc, bool av, bool bv, dim3 grid,
                                    hipStream_t s, const GemmArgs& a) {
    if (a_kc && b_kc) fx_gemm_dispatch_vec<BM, BN, true, true>(av, bv, grid, s, a);
    else if (a_kc) fx_gemm_dispatch_vec<BM, BN, true, false>(av, bv, grid, s, a);
    else if (b_kc) fx_gemm_dispatch_vec<BM, BN, false, true>(av, bv, grid, s, a);
    else fx_gemm_dispatch_vec<BM, BN, false, false>(av, bv, grid, s, a);
}

// Validation, K split and tile choice of one GEMM: fills `a`, bm, bn (M == 0 || N == 0: nothing to do,
// a.M stays 0).
static int fx_gemm_prepare(int32_t transa, int32_t transb, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                           int64_t ldc, const fx_gemm_epilogue* epi_host, int32_t split_k,
                           float* workspace, GemmArgs& a, int& bm_out, int& bn_out) {
    memset(&a, 0, sizeof(a));
    FX_CHECK_ARG(M >= 0 && N >= 0 && K >= 0, "fx_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return FX_OK;
    FX_CHECK_ARG(A && B && C, "fx_gemm_f32: null matrix");
    FX_CHECK_ARG(lda >= (transa ? M : K) && ldb >= (transb ? K : N) && ldc >= N,
                 "fx_gemm_f32: leading dimension too small (lda=%lld ldb=%lld ldc=%lld)",
                 (long long)lda, (long long)ldb, (long long)ldc);
    if (split_k < 1) split_k = 1;
    FX_CHECK_ARG(split_k == 1 || workspace, "fx_gemm_f32: split_k > 1 needs a workspace");
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K;
    if (epi_host) a.epi = *epi_host;
    // K per split, rounded up to the k-tile so every split starts on a 16-B boundary
    int64_t kc = fx_ceil_div(fx_ceil_div(K, split_k), FX_BK) * FX_BK;
    if (kc < FX_BK) kc = FX_BK;
    split_k = (int32_t)fx_ceil_div(K > 0 ? K : 1, kc);
    a.k_chunk = kc;
    a.split_k = split_k;
    a.ws = workspace;
    // tile shape: the largest one that still gives ~2 workgroups per CU (256 CUs)
    int bm = 128, bn = 128;
    {
        static const int forced = []() {   // FX_GEMM_TILE=128x128|128x64|64x64 (experiments)
            const char* e = getenv("FX_GEMM_TILE");
            if (!e) return 0;
            if (!strcmp(e, "128x128")) return 1;
            if (!strcmp(e, "128x64")) return 2;
            if (!strcmp(e, "64x64")) return 3;
            return 0;
        }();
        // Measured on MI355X (profiles/r01_gemm_pipe.txt).  Bare GEMMs favour the big tile (4096^3:
        // 134 TFLOP/s at 128x128 vs 116 at 64x64), but inside the training step every GEMM has an
        // epilogue that touches 16-32 MB (bias+ReLU, ReLU mask, split-K slabs): with one 128x128
        // workgroup per CU all epilogues run at the same time and nothing covers them; with 64x64
        // tiles three workgroups share a CU and one's epilogue hides under the others' MFMAs.
        // Whole DeepFM step: 1.20 ms (64x64) vs 1.22 (128x64) vs 1.24 (128x128 where it fits).
        // So: big tiles only when there are >= 4 waves of them (epilogues then overlap anyway).
        const int64_t t128 = fx_ceil_div(M, 128) * fx_ceil_div(N, 128) * split_k;
        const int64_t t12864 = fx_ceil_div(M, 128) * fx_ceil_div(N, 64) * split_k;
        if (forced == 2) { bn = 64; }
        else if (forced == 3) { bm = 64; bn = 64; }
        else if (forced == 0 && N <= 64) {
            bn = 64;   // e.g. the DIN attention MLP: 204800 x 64 x 64 — do not pad N to 128
        } else if (forced == 0 && !fx_gemm_pipe_mode()) {
            if (t128 < 448) { bm = 64; bn = 64; }      // the unpipelined kernel's rule
        // (round 3: with the 16-byte epilogue a 128x128 workgroup per CU ties with four 64x64 ones on
        // some boxes — 71.3 vs 71.4 us for 4096x1024x1024 — and loses 4 % on others — 78.2 vs 75.2 us,
        // whole DeepFM step 1.043 vs 1.026 ms, profiles/r03_gemm_lab_h.txt: the single launches keep 64x64)
        } else if (forced == 0 && t128 < 1024) {
            if (t12864 >= 2048) bn = 64;
            else { bm = 64; bn = 64; }
        }
    }
    a.tiles_m = (int32_t)fx_ceil_div(M, bm);
    a.tiles_n = (int32_t)fx_ceil_div(N, bn);
    {
        static const int edge_plain = []() {   // FX_GEMM_EDGE_PLAIN=0: masked bodies on edge tiles (A/B)
            const char* e = getenv("FX_GEMM_EDGE_PLAIN");
            return e ? atoi(e) : 1;
        }();
        a.edge_plain = edge_plain;
    }
#ifdef FX_GEMM_LAB
    a.trace = fx_gemm_lab_trace;
#endif
    bm_out = bm;
    bn_out = bn;
    return FX_OK;
}

// operand alignment / offset range the pipelined kernel needs
static bool fx_gemm_pipe_ok(int32_t transa, int32_t transb, const GemmArgs& a) {
    const bool a_kc = !transa, b_kc = transb != 0;
    const bool a_al = (a.lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.A) & 15) == 0);
    const bool b_al = (a.ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.B) & 15) == 0);
    const bool av = a_al && (a_kc ? (a.K % 4 == 0) : (a.M % 4 == 0));
    const bool bv = b_al && (b_kc ? (a.K % 4 == 0) : (a.N % 4 == 0));
    const bool small_offsets = (transa ? a.K * a.lda : a.M * a.lda) < (int64_t)0x3FFFFFF0 &&
                               (transb ? a.N * a.ldb : a.K * a.ldb) < (int64_t)0x3FFFFFF0;
    return fx_gemm_pipe_mode() && av && bv && small_offsets && a.k_chunk >= 4;
}

// The split-bf16 kernels (fx_gemm_x6.hip) take a prepared problem when its operands satisfy the pipelined
// kernel's alignment rules and the 16-byte epilogue's, the output has at least a few 128x128 tiles per slab
// and a fused row sum is asked of a row-contiguous op(A) only (the weight gradients' dZ^T).  Measured on one
// box against the fp32-MFMA kernels (profiles/r05_gemm_x6s_lab_a.txt): 4096x1024x1024 forward 50.6 vs 86.6 us,
// dX 52.4 vs 88.5, dW (4 slabs) 57.3 vs 91.7, 4096x624x624 29.3 vs 43.0; a 256x256x128 product loses
// (8.4 vs 7.2 us): small outputs stay on the fp32 kernels' 64x64 tiles.
#define FX_X6_MIN_WGS 96
static bool fx_gemm_x6_shape(int64_t M, int64_t N, int64_t K) {
    return fx_gemm_x6_enabled() && M >= 128 && N >= 128 && K >= 64;
}

static bool fx_gemm_x6_ok(int32_t transa, int32_t transb, const GemmArgs& a) {
    if (!fx_gemm_x6_shape(a.M, a.N, a.K)) return false;
    if (!fx_gemm_pipe_ok(transa, transb, a) || !fx_gemm_tr_ok(a)) return false;
    if (a.epi.rowsum && !transa) return false;
    // one 8-wave workgroup per CU at ~1.7x the fp32 kernels' per-CU rate against 64x64 tiles that fill the chip
    // with a quarter of the output.  Measured on one box (profiles/r05_gemm_x6_tile_threshold.txt, K = 1024
    // forward with bias + ReLU): 128 tiles 40.5 vs 48.6 us, 96 tiles 38.6 vs 46.7, 64 tiles 36.6 vs 30.8 —
    // the split-bf16 kernels take a launch from 96 workgroups up
    return fx_ceil_div(a.M, 128) * fx_ceil_div(a.N, 128) * a.split_k >= FX_X6_MIN_WGS;
}

// K slabs of a problem on the x6 kernels: about 1024 deep (fx_multi_plan's rule), within the caller's cap
static int32_t fx_splitk_rule_x6(int64_t K, int32_t cap) {
    if (cap <= 1) return 1;
    int64_t sk = (K + 512) / 1024;
    if (sk > cap) sk = cap;
    return (int32_t)(sk < 1 ? 1 : sk);
}

extern "C" int fx_gemm_f32(int32_t transa, int32_t transb, int64_t M, int64_t N, int64_t K,
                           const float* A, int64_t lda, const float* B, int64_t ldb, float* C,
                           int64_t ldc, const fx_gemm_epilogue* epi_host, int32_t split_k,
                           float* workspace, fx_stream_t stream) {
    GemmArgs a;
    int bm = 0, bn = 0;
    const int rc0 = fx_gemm_prepare(transa, transb, M, N, K, A, lda, B, ldb, C, ldc, epi_host, split_k,
                                    workspace, a, bm, bn);
    if (rc0 != FX_OK) return rc0;
    if (M == 0 || N == 0) return FX_OK;
    split_k = a.split_k;
    const int64_t kc = a.k_chunk;
    hipStream_t s = fx_hip_stream(stream);
    const bool want_rowsum = a.epi.rowsum != nullptr;
    FX_CHECK_ARG(!want_rowsum || (K > 8 && !(N <= 4 && !transa)),
                 "fx_gemm_f32: epilogue.rowsum is not available on the K<=8 / N<=4 skinny paths");
    if (K <= 8) {
        a.split_k = 1;
        if (N % 4 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 &&
            M * (N / 4) < ((int64_t)1 << 31)) {
            int64_t blocks = fx_ceil_div(M * (N / 4), 256);
            if (blocks > 16384) blocks = 16384;
            hipLaunchKernelGGL(k_gemm_small_k_v4, dim3((unsigned)blocks), dim3(256), 0, s, a,
                               (int)(transa != 0), (int)(transb != 0));
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
        int64_t blocks = fx_ceil_div(M * N, 256);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_gemm_small_k, dim3((unsigned)blocks), dim3(256), 0, s, a,
                           (int)(transa != 0), (int)(transb != 0));
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    if (N <= 4 && !transa) {
        a.split_k = 1;
        const bool v4 = K <= 256 && K % 4 == 0 && lda % 4 == 0 &&
                        ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
        if (v4) {
            int lpr_log2 = 0;
            while ((4 << lpr_log2) < K) ++lpr_log2;                 // lanes per row = pow2 >= K/4
            const int rpw = 64 >> lpr_log2;
            int64_t blocks = fx_ceil_div(M, 4 * rpw);
            if (blocks > 16384) blocks = 16384;
            dim3 g((unsigned)blocks);
            const int tbi = (int)(transb != 0);
            switch (lpr_log2) {
                case 0: hipLaunchKernelGGL(k_gemm_small_n_v4<0>, g, dim3(256), 0, s, a, tbi); break;
                case 1: hipLaunchKernelGGL(k_gemm_small_n_v4<1>, g, dim3(256), 0, s, a, tbi); break;
                case 2: hipLaunchKernelGGL(k_gemm_small_n_v4<2>, g, dim3(256), 0, s, a, tbi); break;
                case 3: hipLaunchKernelGGL(k_gemm_small_n_v4<3>, g, dim3(256), 0, s, a, tbi); break;
                case 4: hipLaunchKernelGGL(k_gemm_small_n_v4<4>, g, dim3(256), 0, s, a, tbi); break;
                case 5: hipLaunchKernelGGL(k_gemm_small_n_v4<5>, g, dim3(256), 0, s, a, tbi); break;
                default: hipLaunchKernelGGL(k_gemm_small_n_v4<6>, g, dim3(256), 0, s, a, tbi); break;
            }
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
        int64_t blocks = fx_ceil_div(M, 4);
        if (blocks > 8192) blocks = 8192;
        if (transb && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0 &&
            ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) & 15) == 0) {
            hipLaunchKernelGGL(k_gemm_small_n_wide, dim3((unsigned)blocks), dim3(256), 0, s, a);
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
        hipLaunchKernelGGL(k_gemm_small_n, dim3((unsigned)blocks), dim3(256), 0, s, a,
                           (int)(transb != 0));
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    if (M <= 4 && transa && !transb && workspace) {
        // finer K split than the MFMA path wants: this kernel is a column-parallel reduction
        int64_t want = split_k > 1 ? split_k : 1;
        int64_t kc2 = fx_ceil_div(K, want);
        if (kc2 < 1) kc2 = 1;
        a.k_chunk = kc2;
        a.split_k = (int32_t)fx_ceil_div(K, kc2);
        const bool v4 = (N >= 16) && (N % 4 == 0) && (ldb % 4 == 0) &&
                        ((reinterpret_cast<uintptr_t>(B) & 15) == 0) &&
                        ((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
        int np_log2 = 0;
        while ((1 << np_log2) < N && np_log2 < 8) ++np_log2;
        if (v4) {
            int cg_log2 = 2;                                        // float4 column groups per block
            while ((4 << cg_log2) < N && cg_log2 < 6) ++cg_log2;
            dim3 g((unsigned)fx_ceil_div(N, 4 << cg_log2), (unsigned)a.split_k);
            switch (cg_log2) {
                case 2: hipLaunchKernelGGL(k_gemm_small_m_v4<2>, g, dim3(256), 0, s, a); break;
                case 3: hipLaunchKernelGGL(k_gemm_small_m_v4<3>, g, dim3(256), 0, s, a); break;
                case 4: hipLaunchKernelGGL(k_gemm_small_m_v4<4>, g, dim3(256), 0, s, a); break;
                case 5: hipLaunchKernelGGL(k_gemm_small_m_v4<5>, g, dim3(256), 0, s, a); break;
                default: hipLaunchKernelGGL(k_gemm_small_m_v4<6>, g, dim3(256), 0, s, a); break;
            }
        } else
            hipLaunchKernelGGL(k_gemm_small_m,
                               dim3((unsigned)fx_ceil_div(N, 1 << np_log2), (unsigned)a.split_k),
                               dim3(256), 0, s, a, np_log2);
        FX_CHECK_LAUNCH();
        fx_launch_splitk_reduce(a, s);
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    const bool a_kc = !transa, b_kc = transb != 0;
    if (fx_gemm_x6_ok(transa, transb, a)) {
        a.tiles_m = (int32_t)fx_ceil_div(M, 128);
        a.tiles_n = (int32_t)fx_ceil_div(N, 128);
        const int rc = fx_gemm_x6_launch(a_kc, b_kc, a, s);
        if (rc != FX_OK) return rc;
        if (split_k > 1) {
            fx_launch_splitk_reduce(a, s);
            FX_CHECK_LAUNCH();
        }
        return FX_OK;
    }
    const bool a_al = (lda % 4 == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
    const bool b_al = (ldb % 4 == 0) && ((reinterpret_cast<uintptr_t>(B) & 15) == 0);
    const bool av = a_al && (a_kc ? (K % 4 == 0) : (M % 4 == 0));
    const bool bv = b_al && (b_kc ? (K % 4 == 0) : (N % 4 == 0));
    dim3 grid((unsigned)((int64_t)a.tiles_m * a.tiles_n), (unsigned)split_k);
    const int pipe_mode = fx_gemm_pipe_mode();
    const bool small_offsets = (transa ? K * lda : M * lda) < (int64_t)0x3FFFFFF0 &&
                               (transb ? N * ldb : K * ldb) < (int64_t)0x3FFFFFF0;   // 32-bit byte offsets
    if (pipe_mode && av && bv && small_offsets && kc >= 4) {
        int rc;
        if (bm == 128 && bn == 128) rc = fx_gemm_dispatch_pipe<128, 128>(a_kc, b_kc, grid, s, a);
        else if (bm == 128) rc = fx_gemm_dispatch_pipe<128, 64>(a_kc, b_kc, grid, s, a);
        else rc = fx_gemm_dispatch_pipe<64, 64>(a_kc, b_kc, grid, s, a);
        if (rc != FX_OK) return rc;
    } else if (bm == 128 && bn == 128) fx_gemm_dispatch_layout<128, 128>(a_kc, b_kc, av, bv, grid, s, a);
    else if (bm == 128) fx_gemm_dispatch_layout<128, 64>(a_kc, b_kc, av, bv, grid, s, a);
    else fx_gemm_dispatch_layout<64, 64>(a_kc, b_kc, av, bv, grid, s, a);
    FX_CHECK_LAUNCH();
    if (split_k > 1) {
        fx_launch_splitk_reduce(a, s);
        FX_CHECK_LAUNCH();
    }
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_gemm_f32_batch: several independent GEMMs in as few launches as possible.
//   multi path  (default for 2 .. FX_MULTI_MAX aligned, non-skinny problems of a backward pass): ONE
//               k_gemm_f32_multi grid on 128-row tiles, two workgroups per CU; per problem the tile
//               (128x128 | 128x64) and — where the caller allows a K split (split_k > 1) — the number of
//               K slabs follow fx_multi_plan's rules, longest workgroups first in the grid;
//   pair path   (FX_GEMM_MULTI=0): the dW / dX pair of round 2 on 64x64 tiles;
//   else        problem by problem.
// ---------------------------------------------------------------------------------------------
struct MultiPlanItem {
    int tile;      // 0: 128x128, 1: 128x64
    int sk;
};

struct MultiShape {
    int64_t M, N, K;
    int splittable;      // 0: no K split; else the LARGEST number of slabs the caller's workspace holds
};

// Tile and K split of every problem of one multi-problem launch.  Rules read off the forced-configuration
// sweeps of round 3 (profiles/r03_gemm_lab_f.txt; a list-scheduling model of the launch was tried first
// and ranked the candidates poorly — 16 % off the best configuration on DCNv2's first layer):
//   * a K split brings the slabs of a weight gradient to ~1024 deep (what the other problems of the
//     launch run: equal workgroup lengths pack best), within the caller's cap;
//   * 128x128 tiles, unless the problem then has fewer workgroups than CUs AND is one of the small
//     products of the launch (< 0.7 of the largest one's flops): 128x64 doubles its workgroups (the
//     624-wide CrossNet products: 100 / 160 -> 200 / 320 workgroups fill the second slot of the CUs
//     beside the deep layer's 256 + 256) — 258 -> 234 us on cross + deep 1024, 201 -> 182 us on the
//     first layer;
//   * returns the launch's workgroup count on 128x128 tiles: below 512 (two per CU) the grid cannot
//     keep both slots busy and the 64x64 pair kernel / single launches win (pair 1024x624: 147 vs 126
//     us, 256x512: 57 vs 30 us, profiles/r03_gemm_lab_g.txt) — the caller falls back.
// FX_MULTI_CFG="tile,sk;tile,sk;..." overrides (experiments).
static int64_t fx_multi_plan(const MultiShape* sh, int n, MultiPlanItem* plan) {
    double fmax = 0.0;
    for (int i = 0; i < n; ++i) {
        const double f = (double)sh[i].M * sh[i].N * sh[i].K;
        if (f > fmax) fmax = f;
    }
    int64_t wg128 = 0;         // workgroups of the launch on 128x128 tiles
    for (int i = 0; i < n; ++i) {
        int sk = 1;
        if (sh[i].splittable > 1) {
            sk = (int)((sh[i].K + 512) / 1024);
            if (sk > sh[i].splittable) sk = sh[i].splittable;
            if (sk < 1) sk = 1;
        }
        const int64_t t128 = fx_ceil_div(sh[i].M, 128) * fx_ceil_div(sh[i].N, 128);
        const double f = (double)sh[i].M * sh[i].N * sh[i].K;
        plan[i].sk = sk;
        plan[i].tile = (sh[i].N <= 64 || (t128 * sk < 256 && f < 0.7 * fmax)) ? 1 : 0;
        wg128 += t128 * sk;
    }
    static const char* forced = getenv("FX_MULTI_CFG");
    if (forced) {
        const char* q = forced;
        for (int i = 0; i < n && *q; ++i) {
            plan[i].tile = atoi(q);
            while (*q && *q != ',' && *q != ';') ++q;
            if (*q == ',') {
                ++q;
                const int sk = atoi(q);
                if (sk >= 1 && sk <= (sh[i].splittable ? sh[i].splittable : 1)) plan[i].sk = sk;
                while (*q && *q != ';') ++q;
            }
            if (*q == ';') ++q;
            if (sh[i].N <= 64) plan[i].tile = 1;
        }
        return 1 << 20;
    }
    return wg128;
}

// split_k of a problem is the LARGEST slab count its workspace holds; the 64x64 paths use the rule of
// round 2 below that cap: about 1024 workgroups, a power of two, K slabs of >= 256
static int32_t fx_splitk_rule64(int64_t M, int64_t N, int64_t K, int32_t cap) {
    if (cap <= 1) return 1;
    const int64_t tiles = fx_ceil_div(M, 64) * fx_ceil_div(N, 64);
    int64_t s = 1;
    if (tiles >= 768) s = 1;
    else if (tiles <= 4) { s = fx_ceil_div(512, tiles); if (s > 256) s = 256; }
    else {
        const double want = 1024.0 / (double)tiles;
        while ((double)s * 1.5 < want) s *= 2;
        if (s > 16) s = 16;
    }
    if (s > K / 256) s = K / 256;
    if (s > cap) s = cap;
    return (int32_t)(s < 1 ? 1 : s);
}

static bool fx_gemm_skinny(const fx_gemm_problem& q) {
    return q.K <= 8 || (q.N <= 4 && !q.transa) || (q.M <= 4 && q.transa && !q.transb && q.workspace);
}

static int fx_gemm_multi_mode() {     // FX_GEMM_MULTI=0: the 64x64 pair / per-problem paths of round 2
    static const int mode = []() {
        const char* e = getenv("FX_GEMM_MULTI");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

// -> FX_OK and *launched = true when the problems went out as one k_gemm_f32_multi grid
// K slabs of the problems of one split-bf16 multi-problem launch.  One 8-wave workgroup per CU: the launch is a
// list-scheduling problem on 256 machines — a workgroup costs a fixed part (pipeline fill + epilogue: ~9 us) plus
// ~1.2 us per k tile, workgroups start in grid order (longest first) as CUs free up — and the slab count of the
// weight gradients is the one free parameter: 1024 x 624 x 4096 (dW) beside 4096 x 624 x 1024 (dX) is 160 + 160
// workgroups of 32 k tiles at 4 slabs — a full round and a quarter-full one, 64 k tiles of wall time — but
// 320 + 160 workgroups that pack into ~48 at 8 slabs.  Every combination of {1,2,3,4,6,8,12,16} slabs (within
// the caller's workspace cap, slabs >= 256 deep) is simulated, plus what its slab reduces cost; the best one is
// kept per shape set (the simulation runs once per distinct launch).
struct FxX6PlanKey {
    int64_t d[FX_MULTI_MAX][4];
    int n;
    bool operator==(const FxX6PlanKey& o) const { return n == o.n && memcmp(d, o.d, sizeof(d)) == 0; }
};

static double fx_x6_makespan(const int64_t* tiles, const int64_t* ktiles, const int32_t* sk, const int* order,
                             int n) {
    double cu[256];
    for (int i = 0; i < 256; ++i) cu[i] = 0.0;
    // CUs as a binary min-heap on their free time (all zero: already a heap)
    auto sift = [&](int i) {
        for (;;) {
            int l = 2 * i + 1, r = l + 1, m = i;
            if (l < 256 && cu[l] < cu[m]) m = l;
            if (r < 256 && cu[r] < cu[m]) m = r;
            if (m == i) return;
            const double t = cu[i]; cu[i] = cu[m]; cu[m] = t;
            i = m;
        }
    };
    double end = 0.0;
    for (int oi = 0; oi < n; ++oi) {
        const int i = order[oi];
        const double len = 9.0 + 1.2 * (double)fx_ceil_div(ktiles[i], sk[i]);
        const int64_t jobs = tiles[i] * sk[i];
        for (int64_t j = 0; j < jobs; ++j) {
            cu[0] += len;
            if (cu[0] > end) end = cu[0];
            sift(0);
        }
    }
    return end;
}

static void fx_x6_plan_splits(const fx_gemm_problem* p, int n, int32_t* sk_out) {
    // (forward batches are launched from the main thread, backward ones from the autograd engine's thread: the
    // cache is locked — ADVICE r5 — and a plan never exceeds this problem's workspace cap, see the clamp below)
    static FxX6PlanKey keys[64];
    static int32_t plans[64][FX_MULTI_MAX];
    static int n_cached = 0, n_next = 0;
    static std::mutex mtx;
    std::lock_guard<std::mutex> lock(mtx);
    FxX6PlanKey key;
    memset(&key, 0, sizeof(key));
    key.n = n;
    int64_t tiles[FX_MULTI_MAX], ktiles[FX_MULTI_MAX];
    int32_t cap[FX_MULTI_MAX];
    for (int i = 0; i < n; ++i) {
        cap[i] = (p[i].split_k > 1 && p[i].workspace) ? p[i].split_k : 1;
        key.d[i][0] = p[i].M; key.d[i][1] = p[i].N; key.d[i][2] = p[i].K; key.d[i][3] = cap[i];
        tiles[i] = fx_ceil_div(p[i].M, 128) * fx_ceil_div(p[i].N, 128);
        ktiles[i] = fx_ceil_div(p[i].K, FX_BK);
    }
    for (int c = 0; c < n_cached; ++c)
        if (keys[c] == key) {
            for (int i = 0; i < n; ++i) sk_out[i] = plans[c][i] <= cap[i] ? plans[c][i] : cap[i];
            return;
        }
    static const bool planner_on = []() {     // FX_X6_PLAN=0: ~1024-deep slabs (round 5's first cut)
        const char* e = getenv("FX_X6_PLAN");
        return !(e && atoi(e) == 0);
    }();
    static const int cand[8] = {1, 2, 3, 4, 6, 8, 12, 16};
    int32_t best[FX_MULTI_MAX], cur[FX_MULTI_MAX];
    for (int i = 0; i < n; ++i) best[i] = cur[i] = fx_splitk_rule_x6(p[i].K, cap[i]);
    if (planner_on) {
        double best_t = 1e30;
        int idx[FX_MULTI_MAX] = {0, 0, 0, 0};
        for (;;) {
            bool ok = true;
            for (int i = 0; i < n && ok; ++i) {
                cur[i] = cap[i] > 1 ? cand[idx[i]] : 1;
                ok = cur[i] <= cap[i] && (cur[i] == 1 || p[i].K / cur[i] >= 256);
            }
            if (ok) {
                int order[FX_MULTI_MAX];
                for (int i = 0; i < n; ++i) order[i] = i;
                for (int a2 = 0; a2 < n; ++a2)
                    for (int b2 = a2 + 1; b2 < n; ++b2) {
                        const double wa = (double)fx_ceil_div(p[order[a2]].K, cur[order[a2]]);
                        const double wb = (double)fx_ceil_div(p[order[b2]].K, cur[order[b2]]);
                        if (wb > wa) { const int t = order[a2]; order[a2] = order[b2]; order[b2] = t; }
                    }
                double t = fx_x6_makespan(tiles, ktiles, cur, order, n);
                for (int i = 0; i < n; ++i)        // the slab reduce launches that follow (k_splitk_reduce_v4)
                    if (cur[i] > 1) t += 2.5 + (double)cur[i] * (double)p[i].M * (double)p[i].N * 4.0 / 5.0e6;
                if (t < best_t - 1e-9) {
                    best_t = t;
                    for (int i = 0; i < n; ++i) best[i] = cur[i];
                }
            }
            int i = 0;
            for (; i < n; ++i) {
                if (cap[i] > 1 && ++idx[i] < 8) break;
                idx[i] = 0;
            }
            if (i == n) break;
        }
    }
    for (int i = 0; i < n; ++i) sk_out[i] = best[i] <= cap[i] ? best[i] : cap[i];
    keys[n_next] = key;                       // (64 shape sets, oldest replaced)
    for (int i = 0; i < n; ++i) plans[n_next][i] = best[i];
    n_next = (n_next + 1) & 63;
    if (n_cached < 64) ++n_cached;
}

// Which slab rule a problem of a batch gets when it is launched on its own: the x6 rule (~1024-deep
// slabs) only if the x6 kernels will really take it — enough 128x128 tiles x slabs, a fused row sum only on a
// row-contiguous op(A); a 256x256 or 512x512 weight gradient with K = 4096 otherwise ran on the fp32 kernels'
// 64x64 tiles with 4 slabs where their own rule gives 16 (a quarter of the chip).  (Operand alignment is checked
// again in fx_gemm_f32; a misaligned problem only loses the better split.)
static bool fx_batch_wants_x6(const fx_gemm_problem& q) {
    if (!fx_gemm_x6_shape(q.M, q.N, q.K)) return false;
    if (q.epilogue && q.epilogue->rowsum && !q.transa) return false;
    const int32_t sk = fx_splitk_rule_x6(q.K, q.workspace ? q.split_k : 1);
    return fx_ceil_div(q.M, 128) * fx_ceil_div(q.N, 128) * sk >= FX_X6_MIN_WGS;
}

// The same on the split-bf16 kernels (one workgroup of 8 waves per CU, 128x128 tiles only): any 2 .. 4 problems
// that all qualify, K-split or not — the tiles of the shorter problems fill the CUs the longest one leaves.
static int fx_gemm_try_multi_x6(const fx_gemm_problem* p, int32_t n, fx_stream_t stream, bool* launched) {
    *launched = false;
    if (n < 2 || n > FX_MULTI_MAX || !fx_gemm_x6_enabled() || !fx_gemm_multi_mode()) return FX_OK;
    for (int i = 0; i < n; ++i) {
        const fx_gemm_problem& q = p[i];
        if (fx_gemm_skinny(q) || !q.A || !q.B || !q.C || !fx_gemm_x6_shape(q.M, q.N, q.K)) return FX_OK;
    }
    MultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    int order[FX_MULTI_MAX];
    int32_t sk[FX_MULTI_MAX];
    double wl[FX_MULTI_MAX];
    fx_x6_plan_splits(p, n, sk);
    for (int i = 0; i < n; ++i) {
        order[i] = i;
        wl[i] = (double)fx_ceil_div(p[i].K, sk[i]) + (sk[i] == 1 ? 1.0 : 0.0);     // longest workgroups first
    }
    for (int a2 = 0; a2 < n; ++a2)
        for (int b2 = a2 + 1; b2 < n; ++b2)
            if (wl[order[b2]] > wl[order[a2]]) { const int t = order[a2]; order[a2] = order[b2]; order[b2] = t; }
    int64_t wgs = 0;
    for (int oi = 0; oi < n; ++oi) {
        const fx_gemm_problem& q = p[order[oi]];
        GemmArgs& a = ma.p[oi];
        int bm = 0, bn = 0;
        const int rc = fx_gemm_prepare(q.transa, q.transb, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C, q.ldc,
                                       q.epilogue, sk[order[oi]], q.workspace, a, bm, bn);
        if (rc != FX_OK) return rc;
        a.tiles_m = (int32_t)fx_ceil_div(q.M, 128);
        a.tiles_n = (int32_t)fx_ceil_div(q.N, 128);
        if (!fx_gemm_pipe_ok(q.transa, q.transb, a) || !fx_gemm_tr_ok(a) || (a.epi.rowsum && !q.transa))
            return FX_OK;
        ma.cfg[oi] = (q.transa ? 0 : 1) | (q.transb ? 2 : 0);
        ma.start[oi] = (int32_t)wgs;
        wgs += (int64_t)a.tiles_m * a.tiles_n * a.split_k;
    }
    ma.start[n] = (int32_t)wgs;
    for (int oi = n + 1; oi <= FX_MULTI_MAX; ++oi) ma.start[oi] = (int32_t)wgs;
    ma.n = n;
    if (wgs < FX_X6_MIN_WGS || wgs > 0x3FFFFFFF) return FX_OK;
    hipStream_t s = fx_hip_stream(stream);
    const int rc = fx_gemm_x6_launch_multi(ma, wgs, s);
    if (rc != FX_OK) return rc;
    fx_launch_splitk_reduces(ma.p, n, s);
    FX_CHECK_LAUNCH();
    *launched = true;
    return FX_OK;
}

static int fx_gemm_try_multi(const fx_gemm_problem* p, int32_t n, fx_stream_t stream, bool* launched) {
    *launched = false;
    if (n < 2 || n > FX_MULTI_MAX || !fx_gemm_multi_mode()) return FX_OK;
    {
        const int rc = fx_gemm_try_multi_x6(p, n, stream, launched);
        if (rc != FX_OK || *launched) return rc;
    }
    // (two forward-type products — no K split anywhere — measured no better as one grid than as two
    // launches with their own tile shapes: 125 vs 122 us for DCNv2's cross + deep forward)
    bool any_split = false;
    for (int i = 0; i < n; ++i) any_split = any_split || (p[i].split_k > 1 && p[i].workspace);
    if (!any_split) return FX_OK;
    MultiShape sh[FX_MULTI_MAX];
    for (int i = 0; i < n; ++i) {
        const fx_gemm_problem& q = p[i];
        if (q.M < 64 || q.N < 16 || q.K < 64 || fx_gemm_skinny(q) || !q.A || !q.B || !q.C) return FX_OK;
        sh[i].M = q.M; sh[i].N = q.N; sh[i].K = q.K;
        sh[i].splittable = (q.split_k > 1 && q.workspace) ? q.split_k : 0;
    }
    MultiPlanItem plan[FX_MULTI_MAX];
    if (fx_multi_plan(sh, n, plan) < 512) return FX_OK;
    MultiArgs ma;
    memset(&ma, 0, sizeof(ma));
    // longest workgroups first
    int order[FX_MULTI_MAX];
    double wl[FX_MULTI_MAX];
    for (int i = 0; i < n; ++i) {
        order[i] = i;
        wl[i] = (double)fx_ceil_div(p[i].K, plan[i].sk) * (plan[i].tile ? 64 : 128) + (plan[i].sk == 1 ? 1.0 : 0.0);
    }
    for (int a2 = 0; a2 < n; ++a2)
        for (int b2 = a2 + 1; b2 < n; ++b2)
            if (wl[order[b2]] > wl[order[a2]]) { const int t = order[a2]; order[a2] = order[b2]; order[b2] = t; }
    int64_t wgs = 0;
    for (int oi = 0; oi < n; ++oi) {
        const fx_gemm_problem& q = p[order[oi]];
        const MultiPlanItem& pl = plan[order[oi]];
        GemmArgs& a = ma.p[oi];
        int bm = 0, bn = 0;
        const int rc = fx_gemm_prepare(q.transa, q.transb, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C,
                                       q.ldc, q.epilogue, pl.sk, q.workspace, a, bm, bn);
        if (rc != FX_OK) return rc;
        bn = pl.tile ? 64 : 128;
        a.tiles_m = (int32_t)fx_ceil_div(q.M, 128);
        a.tiles_n = (int32_t)fx_ceil_div(q.N, bn);
        if (!fx_gemm_pipe_ok(q.transa, q.transb, a) || !fx_gemm_tr_ok(a)) return FX_OK;
        ma.cfg[oi] = (q.transa ? 0 : 1) | (q.transb ? 2 : 0) | (pl.tile ? 4 : 0);
        ma.start[oi] = (int32_t)wgs;
        wgs += (int64_t)a.tiles_m * a.tiles_n * a.split_k;
    }
    ma.start[n] = (int32_t)wgs;
    for (int oi = n + 1; oi <= FX_MULTI_MAX; ++oi) ma.start[oi] = (int32_t)wgs;
    ma.n = n;
    if (wgs <= 0 || wgs > 0x3FFFFFFF) return FX_OK;
    hipStream_t s = fx_hip_stream(stream);
    hipLaunchKernelGGL(k_gemm_f32_multi, dim3((unsigned)wgs), dim3(256), 0, s, ma);
    FX_CHECK_LAUNCH();
    fx_launch_splitk_reduces(ma.p, n, s);
    FX_CHECK_LAUNCH();
    *launched = true;
    return FX_OK;
}

extern "C" int fx_gemm_f32_batch(const fx_gemm_problem* p, int32_t n, fx_stream_t stream) {
    FX_CHECK_ARG(n >= 0 && (n == 0 || p), "fx_gemm_f32_batch: bad problem list");
    static const bool pair_on = []() {   // FX_GEMM_PAIR=0: always problem by problem (A/B runs)
        const char* e = getenv("FX_GEMM_PAIR");
        return !(e && atoi(e) == 0);
    }();
    if (pair_on) {
        bool launched = false;
        const int rc = fx_gemm_try_multi(p, n, stream, &launched);
        if (rc != FX_OK) return rc;
        if (launched) return FX_OK;
    }
    if (n == 2 && pair_on && p[0].transa && !p[0].transb && !p[1].transa &&
        !p[1].transb) {
        GemmArgs a[2];
        int bm[2], bn[2];
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) {
            const fx_gemm_problem& q = p[i];
            const int rc = fx_gemm_prepare(q.transa, q.transb, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb,
                                           q.C, q.ldc, q.epilogue,
                                           fx_gemm_skinny(q) ? q.split_k
                                                             : fx_splitk_rule64(q.M, q.N, q.K, q.split_k),
                                           q.workspace, a[i], bm[i], bn[i]);
            if (rc != FX_OK) return rc;
            // (the pair kernel is the 64x64 build: force that tile whatever the single-GEMM rule says)
            a[i].tiles_m = (int32_t)fx_ceil_div(q.M, 64);
            a[i].tiles_n = (int32_t)fx_ceil_div(q.N, 64);
            ok = q.M > 0 && q.N > 0 && !fx_gemm_skinny(q) && fx_gemm_pipe_ok(q.transa, q.transb, a[i]);
        }
        if (ok) {
            hipStream_t s = fx_hip_stream(stream);
            const int64_t wgs = (int64_t)a[0].tiles_m * a[0].tiles_n * a[0].split_k +
                                (int64_t)a[1].tiles_m * a[1].tiles_n * a[1].split_k;
            if (fx_gemm_tr_ok(a[0]) && fx_gemm_tr_ok(a[1]))
                hipLaunchKernelGGL((k_gemm_f32_pair<64, 64, false, false, true, false, 4, true>),
                                   dim3((unsigned)wgs), dim3(256), 0, s, a[0], a[1]);
            else
                hipLaunchKernelGGL((k_gemm_f32_pair<64, 64, false, false, true, false, 4, false>),
                                   dim3((unsigned)wgs), dim3(256), 0, s, a[0], a[1]);
            FX_CHECK_LAUNCH();
            for (int i = 0; i < 2; ++i)
                if (a[i].split_k > 1) {
                    fx_launch_splitk_reduce(a[i], s);
                    FX_CHECK_LAUNCH();
                }
            return FX_OK;
        }
    }
    // The two gradients of a Linear(hidden -> 1) head in one pass over the hidden activations
    // (k_head_bwd_v4).  FX_HEAD_FUSE=0: the two skinny launches.
    static const bool head_on = []() {
        const char* e = getenv("FX_HEAD_FUSE");
        return !(e && atoi(e) == 0);
    }();
    if (n == 2 && head_on && p[0].transa && !p[0].transb && p[0].M == 1 && p[0].workspace && p[0].split_k >= 1 &&
        !p[1].transa && !p[1].transb && p[1].K == 1 && p[1].M == p[0].K && p[1].N == p[0].N &&
        p[1].A == p[0].A && p[1].lda == p[0].lda && p[0].N >= 16 && p[0].N % 4 == 0) {
        const fx_gemm_problem& q0 = p[0];
        const fx_gemm_problem& q1 = p[1];
        const fx_gemm_epilogue* e1 = q1.epilogue;
        const bool plain = !e1 || (!e1->bias && !e1->zout && e1->act == 0 && !e1->mul && !e1->add && !e1->rowsum &&
                                   (!e1->mask || (e1->mask == q0.B && e1->ldmask == q0.ldb)));
        const bool al = q0.ldb % 4 == 0 && q1.ldc % 4 == 0 &&
                        (((uintptr_t)q0.B | (uintptr_t)q0.workspace | (uintptr_t)q1.C | (uintptr_t)q1.B) & 15) == 0;
        if (plain && al && q0.B && q0.C && q1.B && q1.C) {
            HeadBwdArgs h;
            int bm = 0, bn = 0;
            const int rc = fx_gemm_prepare(q0.transa, q0.transb, q0.M, q0.N, q0.K, q0.A, q0.lda, q0.B, q0.ldb,
                                           q0.C, q0.ldc, q0.epilogue, q0.split_k, q0.workspace, h.dw, bm, bn);
            if (rc != FX_OK) return rc;
            // (the K split of the skinny weight-gradient kernel, derived exactly as fx_gemm_f32 does — from
            // the slab count fx_gemm_prepare settled on — so that both paths sum the same slabs)
            const int64_t want = h.dw.split_k > 1 ? h.dw.split_k : 1;
            int64_t kc2 = fx_ceil_div(q0.K, want);
            if (kc2 < 1) kc2 = 1;
            h.dw.k_chunk = kc2;
            h.dw.split_k = (int32_t)fx_ceil_div(q0.K, kc2);
            h.dx = q1.C;
            h.ldx = q1.ldc;
            h.w = q1.B;
            h.use_mask = (e1 && e1->mask) ? 1 : 0;
            int cg_log2 = 2;
            while ((4 << cg_log2) < q0.N && cg_log2 < 6) ++cg_log2;
            hipStream_t s = fx_hip_stream(stream);
            dim3 g((unsigned)fx_ceil_div(q0.N, 4 << cg_log2), (unsigned)h.dw.split_k);
            switch (cg_log2) {
                case 2: hipLaunchKernelGGL(k_head_bwd_v4<2>, g, dim3(256), 0, s, h); break;
                case 3: hipLaunchKernelGGL(k_head_bwd_v4<3>, g, dim3(256), 0, s, h); break;
                case 4: hipLaunchKernelGGL(k_head_bwd_v4<4>, g, dim3(256), 0, s, h); break;
                case 5: hipLaunchKernelGGL(k_head_bwd_v4<5>, g, dim3(256), 0, s, h); break;
                default: hipLaunchKernelGGL(k_head_bwd_v4<6>, g, dim3(256), 0, s, h); break;
            }
            FX_CHECK_LAUNCH();
            fx_launch_splitk_reduce(h.dw, s);
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
    }
    // Two FORWARD products (x W^T: the cross layer and the deep layer of one DCNv2 depth) in one grid of
    // 64x64 tiles, the longer problem's tiles first: the 640 tiles of the 624-wide cross product leave
    // 384 of the 1024 resident slots empty on their own and all reach their (four-operand) epilogue
    // together; behind the deep layer's 1024 tiles they fill slots as those retire.  FX_GEMM_FWDPAIR=0:
    // problem by problem.
    static const bool fwdpair_on = []() {
        const char* e = getenv("FX_GEMM_FWDPAIR");
        return !(e && atoi(e) == 0);
    }();
    if (n == 2 && pair_on && fwdpair_on && !p[0].transa && p[0].transb && !p[1].transa && p[1].transb) {
        GemmArgs a[2];
        int bm[2], bn[2];
        bool ok = true;
        for (int i = 0; i < 2 && ok; ++i) {
            const fx_gemm_problem& q = p[i];
            ok = q.M > 0 && q.N > 0 && !fx_gemm_skinny(q);
            if (!ok) break;
            const int rc = fx_gemm_prepare(q.transa, q.transb, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb,
                                           q.C, q.ldc, q.epilogue, 1, q.workspace, a[i], bm[i], bn[i]);
            if (rc != FX_OK) return rc;
            a[i].tiles_m = (int32_t)fx_ceil_div(q.M, 64);
            a[i].tiles_n = (int32_t)fx_ceil_div(q.N, 64);
            ok = a[i].split_k == 1 && fx_gemm_pipe_ok(q.transa, q.transb, a[i]) && fx_gemm_tr_ok(a[i]);
        }
        if (ok) {
            const int64_t t0 = (int64_t)a[0].tiles_m * a[0].tiles_n, t1 = (int64_t)a[1].tiles_m * a[1].tiles_n;
            const bool swap = (double)p[1].N * p[1].K > (double)p[0].N * p[0].K;   // more work per row first
            // (alternating the two problems' tiles in proportion to their counts instead — every CU
            // starting with a mix of long and short tiles — measured 1.490 vs 1.451 ms: worse than two
            // launches; the longer problem first it is)
            hipLaunchKernelGGL((k_gemm_f32_pair<64, 64, true, true, true, true, 4, true>),
                               dim3((unsigned)(t0 + t1)), dim3(256), 0, fx_hip_stream(stream),
                               swap ? a[1] : a[0], swap ? a[0] : a[1]);
            FX_CHECK_LAUNCH();
            return FX_OK;
        }
    }
    for (int i = 0; i < n; ++i) {
        const fx_gemm_problem& q = p[i];
        const int rc = fx_gemm_f32(q.transa, q.transb, q.M, q.N, q.K, q.A, q.lda, q.B, q.ldb, q.C,
                                   q.ldc, q.epilogue,
                                   fx_gemm_skinny(q) ? q.split_k
                                   : fx_batch_wants_x6(q) ? fx_splitk_rule_x6(q.K, q.workspace ? q.split_k : 1)
                                                          : fx_splitk_rule64(q.M, q.N, q.K, q.split_k),
                                   q.workspace, stream);
        if (rc != FX_OK) return rc;
    }
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// column sums (bias gradient), two deterministic stages
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_colsum_stage1(const float* X, int64_t ldx, int64_t M,
                                                       int64_t N, int64_t rows_per_chunk,
                                                       float* ws) {
    __shared__ float red[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + tx;
    const int64_t mb = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    float acc = 0.f;
    if (n < N)
        for (int64_t m = mb + ty; m < me; m += 4) acc += X[m * ldx + n];
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && n < N)
        ws[(int64_t)blockIdx.y * N + n] = (red[tx] + red[tx + 64]) + (red[tx + 128] + red[tx + 192]);
}

// vectorised variant: a thread owns 4 adjacent columns (N % 4 == 0, 16-B aligned rows)
__global__ __launch_bounds__(256) void k_colsum_stage1_v4(const float* X, int64_t ldx, int64_t M,
                                                          int64_t N, int64_t rows_per_chunk,
                                                          float* ws) {
    __shared__ float4 red[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n = ((int64_t)blockIdx.x * 64 + tx) * 4;
    const int64_t mb = (int64_t)blockIdx.y * rows_per_chunk;
    const int64_t me = (mb + rows_per_chunk < M) ? mb + rows_per_chunk : M;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N)
        for (int64_t m = mb + ty; m < me; m += 4) {
            const float4 v = *reinterpret_cast<const float4*>(X + m * ldx + n);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    red[threadIdx.x] = acc;
    __syncthreads();
    if (ty == 0 && n < N) {
        const float4 a0 = red[tx], a1 = red[tx + 64], a2 = red[tx + 128], a3 = red[tx + 192];
        float4 r;
        r.x = (a0.x + a1.x) + (a2.x + a3.x);
        r.y = (a0.y + a1.y) + (a2.y + a3.y);
        r.z = (a0.z + a1.z) + (a2.z + a3.z);
        r.w = (a0.w + a1.w) + (a2.w + a3.w);
        *reinterpret_cast<float4*>(ws + (int64_t)blockIdx.y * N + n) = r;
    }
}

__global__ __launch_bounds__(256) void k_colsum_stage2(const float* ws, int64_t N, int chunks,
                                                       float* out) {
    __shared__ float red[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int64_t n = (int64_t)blockIdx.x * 64 + tx;
    float s = 0.f;
    if (n < N)
        for (int c = ty; c < chunks; c += 4) s += ws[(int64_t)c * N + n];
    red[threadIdx.x] = s;
    __syncthreads();
    if (ty == 0 && n < N) out[n] = (red[tx] + red[tx + 64]) + (red[tx + 128] + red[tx + 192]);
}

extern "C" int fx_colsum(const float* X, int64_t ldx, int64_t M, int64_t N, float* out,
                         float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(M >= 0 && N >= 0, "fx_colsum: negative dimension");
    if (N == 0) return FX_OK;
    FX_CHECK_ARG(X && out && workspace, "fx_colsum: null pointer");
    hipStream_t s = fx_hip_stream(stream);
    const int64_t rpc = fx_ceil_div(M > 0 ? M : 1, FX_COLSUM_CHUNKS);
    const bool vec = (N % 4 == 0) && (ldx % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(X) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(workspace) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(k_colsum_stage1_v4, dim3((unsigned)fx_ceil_div(N, 256), FX_COLSUM_CHUNKS),
                           dim3(256), 0, s, X, ldx, M, N, rpc, workspace);
    else
        hipLaunchKernelGGL(k_colsum_stage1, dim3((unsigned)fx_ceil_div(N, 64), FX_COLSUM_CHUNKS),
                           dim3(256), 0, s, X, ldx, M, N, rpc, workspace);
    FX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_colsum_stage2, dim3((unsigned)fx_ceil_div(N, 64)), dim3(256), 0, s,
                       workspace, N, (int)FX_COLSUM_CHUNKS, out);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// relu backward mask (only for a tower whose last layer is activated)
// ---------------------------------------------------------------------------------------------
// (the incoming gradient may be a column slice of a wider tensor — the backward of the torch.cat that
// joins the towers' outputs hands out strided views: read in place through its row stride instead of
// a .contiguous() copy first)
__global__ __launch_bounds__(256) void k_mask_mul(const float* dy, int64_t dy_ld, const float* y,
                                                  int64_t y_ld, float* out, uint32_t n, uint32_t cols) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t r = i / cols, c = i - r * cols;
        const float d = dy[(int64_t)r * dy_ld + c];
        out[i] = y[(int64_t)r * y_ld + c] > 0.f ? d : 0.f;
    }
}

extern "C" int fx_mask_mul(const float* dy, int64_t dy_ld, const float* y, int64_t y_ld, float* out,
                           int64_t rows, int64_t cols, fx_stream_t stream) {
    const int64_t n = rows * cols;
    if (n <= 0) return FX_OK;
    FX_CHECK_ARG(dy && y && out && dy_ld >= cols && y_ld >= cols, "fx_mask_mul: bad arguments");
    // (32-bit grid-stride counter: i += gridDim.x * 256 must not wrap)
    FX_CHECK_ARG(n < ((int64_t)1 << 31), "fx_mask_mul: more than 2^31 elements");
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_mask_mul, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), dy,
                       dy_ld, y, y_ld, out, (uint32_t)n, (uint32_t)cols);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// CrossNetV2 backward glue
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cross_bwd_prep(const float* dxn, int64_t dxn_ld,
                                                        const float* x0, const float* z, float* t,
                                                        float* dx0, uint32_t n, uint32_t cols,
                                                        int init, int add_dxn) {
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
        const uint32_t r = i / cols;
        const float d = dxn[(int64_t)r * dxn_ld + (i - r * cols)];
        t[i] = d * x0[i];
        float term = d * z[i];
        if (add_dxn) term += d;
        dx0[i] = init ? term : dx0[i] + term;
    }
}

extern "C" int fx_cross_bwd_prep(const float* dxn, int64_t dxn_ld, const float* x0, const float* z,
                                 float* t, float* dx0, int64_t rows, int64_t cols, int32_t init,
                                 int32_t add_dxn, fx_stream_t stream) {
    const int64_t n = rows * cols;
    if (n <= 0) return FX_OK;
    FX_CHECK_ARG(dxn && x0 && z && t && dx0 && dxn_ld >= cols, "fx_cross_bwd_prep: bad arguments");
    FX_CHECK_ARG(n < ((int64_t)1 << 31), "fx_cross_bwd_prep: more than 2^31 elements");
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_cross_bwd_prep, dim3((unsigned)blocks), dim3(256), 0,
                       fx_hip_stream(stream), dxn, dxn_ld, x0, z, t, dx0, (uint32_t)n,
                       (uint32_t)cols, (int)init, (int)add_dxn);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// sigmoid + binary cross entropy (mean) + dloss/dlogit, one workgroup, fixed-order reduction
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sigmoid_bce(const float* logit, const float* y,
                                                      int64_t B, float* prob, float* loss,
                                                      float* dlogit) {
    __shared__ float red[1024];
    float acc = 0.f;
    const float invB = 1.f / (float)B;
    for (int64_t i = threadIdx.x; i < B; i += 1024) {
        const float x = logit[i];
        const float p = 1.f / (1.f + expf(-x));  // torch.sigmoid
        if (prob) prob[i] = p;
        if (!y) continue;  // activation only
        const float t = y[i];
        // F.binary_cross_entropy clamps each log term at -100
        const float lp = fmaxf(logf(p), -100.f);
        const float lq = fmaxf(logf(1.f - p), -100.f);
        acc += -(t * lp + (1.f - t) * lq);
        if (dlogit) {
            // binary_cross_entropy_backward: (p - t) / max((1 - p) * p, 1e-12) * grad, then
            // sigmoid_backward: * p * (1 - p)
            const float dp = (p - t) / fmaxf((1.f - p) * p, 1e-12f) * invB;
            dlogit[i] = dp * ((1.f - p) * p);
        }
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && loss) *loss = red[0] * invB;
}

extern "C" int fx_sigmoid_bce(const float* logit, const float* y, int64_t B, float* prob,
                              float* loss, float* dlogit, fx_stream_t stream) {
    FX_CHECK_ARG(B > 0, "fx_sigmoid_bce: B must be positive");
    FX_CHECK_ARG(logit, "fx_sigmoid_bce: null logit");
    FX_CHECK_ARG(y || (!loss && !dlogit), "fx_sigmoid_bce: loss/dlogit need labels");
    hipLaunchKernelGGL(k_sigmoid_bce, dim3(1), dim3(1024), 0, fx_hip_stream(stream), logit, y, B,
                       prob, loss, dlogit);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// The training step's last mile in one pass over the top hidden layer (round 4): the Linear(K -> 1)
// head forward (+ the term added to the logit), sigmoid + BCE, and the head's backward — dlogit, the
// input gradient dz[m, :] = dlogit[m] w[:] (with the ReLU mask of the hidden layer: h IS the mask; from a
// column on when only the tail of h went through a ReLU — DCNv2's [cross | deep] head input) and
// the slabs of dW = sum_m dlogit[m] h[m, :], db = sum_m dlogit[m], loss = mean_m bce_m.  It replaces
// k_gemm_small_n_wide + k_sigmoid_bce + k_head_bwd_v4 (h streamed twice, three launch boundaries) by
// one launch; k_head_reduce then adds the G slabs in a fixed order (deterministic) instead of
// k_splitk_reduce_wide.  One wave per row, a lane holds the row's float4s k = 4 lane + 256 u (u < NU):
// the dot product is k_gemm_small_n_wide's fmaf chain + fx_wave_sum, so the logit is bit for bit the
// unfused forward's (evaluate / predict run that one); dlogit is k_sigmoid_bce's expression.
// ---------------------------------------------------------------------------------------------
struct HeadTrainArgs {
    const float* h;       // [M, K] hidden activations (row stride ldh)
    int64_t ldh;
    const float* w;       // [K]
    const float* bias;    // [1] or null
    const float* add;     // [M] (stride ldadd) or null: added to the logit after the bias
    int64_t ldadd;
    const float* y;       // [M] labels
    float* logit;         // [M]
    float* dlogit;        // [M]
    float* dz;            // [M, K] (row stride lddz) or null
    int64_t lddz;
    float* ws;            // [G, K] dW slabs | [G] db partials | [G] loss partials
    int64_t M, K;
    float root_scale;     // the root gradient of loss.backward() (1, or 1 / world when sharded)
    int32_t mask_from;    // < 0: no mask; else dz[m, k] = 0 where h[m, k] <= 0 for k >= mask_from (4 | mask_from)
};

template <int NU>
__global__ __launch_bounds__(256) void k_head_train(HeadTrainArgs a) {
    __shared__ float red[4][NU * 256];
    __shared__ float red2[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t G = gridDim.x;
    const float invB = 1.f / (float)a.M;
    float4 wv[NU], acc[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int64_t k = (int64_t)lane * 4 + 256 * u;
        wv[u] = k < a.K ? *reinterpret_cast<const float4*>(a.w + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float b0 = a.bias ? a.bias[0] : 0.f;
    float dbs = 0.f, ls = 0.f;
    for (int64_t m = (int64_t)blockIdx.x * 4 + wave; m < a.M; m += G * 4) {
        const float* row = a.h + m * a.ldh;
        float4 x[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int64_t k = (int64_t)lane * 4 + 256 * u;
            x[u] = k < a.K ? *reinterpret_cast<const float4*>(row + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float t = a.y[m];
        const float ad = a.add ? a.add[m * a.ldadd] : 0.f;
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            if ((int64_t)lane * 4 + 256 * u < a.K) {      // (k_gemm_small_n_wide adds nothing past K either)
                s = fmaf(x[u].x, wv[u].x, s);
                s = fmaf(x[u].y, wv[u].y, s);
                s = fmaf(x[u].z, wv[u].z, s);
                s = fmaf(x[u].w, wv[u].w, s);
            }
        }
        float z = fx_wave_sum(s);
        if (a.bias) z += b0;
        if (a.add) z += ad;
        const float p = 1.f / (1.f + expf(-z));
        const float lp = fmaxf(logf(p), -100.f);
        const float lq = fmaxf(logf(1.f - p), -100.f);
        ls += -(t * lp + (1.f - t) * lq);
        const float dp = (p - t) / fmaxf((1.f - p) * p, 1e-12f) * invB;
        float d = dp * ((1.f - p) * p);
        if (a.root_scale != 1.f) d *= a.root_scale;
        dbs += d;
        if (lane == 0) {
            a.logit[m] = z;
            a.dlogit[m] = d;
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int64_t k = (int64_t)lane * 4 + 256 * u;
            acc[u].x = fmaf(d, x[u].x, acc[u].x);
            acc[u].y = fmaf(d, x[u].y, acc[u].y);
            acc[u].z = fmaf(d, x[u].z, acc[u].z);
            acc[u].w = fmaf(d, x[u].w, acc[u].w);
            if (a.dz && k < a.K) {
                float4 o = make_float4(d * wv[u].x, d * wv[u].y, d * wv[u].z, d * wv[u].w);
                if (a.mask_from >= 0 && k >= a.mask_from) {
                    o.x = x[u].x > 0.f ? o.x : 0.f;
                    o.y = x[u].y > 0.f ? o.y : 0.f;
                    o.z = x[u].z > 0.f ? o.z : 0.f;
                    o.w = x[u].w > 0.f ? o.w : 0.f;
                }
                *reinterpret_cast<float4*>(a.dz + m * a.lddz + k) = o;
            }
        }
    }
    // the four waves' partial sums, added in wave order
#pragma unroll
    for (int u = 0; u < NU; ++u)
        *reinterpret_cast<float4*>(&red[wave][lane * 4 + 256 * u]) = acc[u];
    if (lane == 0) {
        red2[wave] = dbs;
        red2[4 + wave] = ls;
    }
    __syncthreads();
    for (int64_t k = threadIdx.x; k < a.K; k += 256)
        a.ws[(int64_t)blockIdx.x * a.K + k] = ((red[0][k] + red[1][k]) + red[2][k]) + red[3][k];
    if (threadIdx.x == 0) {
        a.ws[G * a.K + blockIdx.x] = ((red2[0] + red2[1]) + red2[2]) + red2[3];
        a.ws[G * a.K + G + blockIdx.x] = ((red2[4] + red2[5]) + red2[6]) + red2[7];
    }
}

// dW[k] = sum over the G slabs (8 elements x 32 slab lanes per workgroup, 8 loads in flight, fixed LDS
// tree); workgroup 0 also adds the G db / loss partials (G <= 256: one per thread, fixed tree).
__global__ __launch_bounds__(256) void k_head_reduce(const float* ws, int64_t G, int64_t K, float invB,
                                                     float* dW, float* db, float* loss) {
    __shared__ float red[256];
    const int ii = threadIdx.x & 7, zi = threadIdx.x >> 3;
    const int64_t k = (int64_t)blockIdx.x * 8 + ii;
    float s = 0.f;
    if (k < K) {
        int64_t z = zi;
        for (; z + 7 * 32 < G; z += 8 * 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(z + u * 32) * K + k];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < G; z += 32) s += ws[z * K + k];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = 16; h > 0; h >>= 1) {
        if (zi < h) red[threadIdx.x] += red[threadIdx.x + h * 8];
        __syncthreads();
    }
    if (zi == 0 && k < K) dW[k] = red[ii];
    if (blockIdx.x == 0) {                       // block-uniform
        for (int which = 0; which < 2; ++which) {
            __syncthreads();
            red[threadIdx.x] = (int64_t)threadIdx.x < G ? ws[G * K + which * G + threadIdx.x] : 0.f;
            __syncthreads();
            for (int h = 128; h > 0; h >>= 1) {
                if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
                __syncthreads();
            }
            if (threadIdx.x == 0) {
                if (which == 0) { if (db) db[0] = red[0]; }
                else if (loss) loss[0] = red[0] * invB;
            }
        }
    }
}

static int64_t fx_head_train_groups(int64_t M) {
    int64_t g = fx_ceil_div(M, 4);
    return g > 256 ? 256 : (g < 1 ? 1 : g);
}

extern "C" int64_t fx_head_train_workspace(int64_t M, int64_t K) {
    return fx_head_train_groups(M) * (K + 2);
}

extern "C" int fx_head_train(const float* h, int64_t ldh, const float* w, const float* bias,
                             const float* add, int64_t ldadd, const float* y, int64_t M, int64_t K,
                             int32_t mask_from, float root_scale, float* logit, float* dlogit, float* dz,
                             int64_t lddz, float* dW, float* db, float* loss, float* workspace,
                             fx_stream_t stream) {
    FX_CHECK_ARG(M > 0 && K > 0, "fx_head_train: M and K must be positive");
    FX_CHECK_ARG(h && w && y && logit && dlogit && dW && workspace, "fx_head_train: null argument");
    // (K <= 8: fx_gemm_f32 takes its one-thread-per-output kernel there — another summation order)
    FX_CHECK_ARG(K % 4 == 0 && K > 8 && K <= 2048, "fx_head_train: K must be a multiple of 4 in (8, 2048] (K=%lld)",
                 (long long)K);
    FX_CHECK_ARG(ldh % 4 == 0 && (reinterpret_cast<uintptr_t>(h) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(w) & 15) == 0 &&
                     (reinterpret_cast<uintptr_t>(workspace) & 15) == 0,
                 "fx_head_train: h / w / workspace must be 16-byte aligned, ldh %% 4 == 0");
    FX_CHECK_ARG(!dz || (lddz % 4 == 0 && (reinterpret_cast<uintptr_t>(dz) & 15) == 0),
                 "fx_head_train: dz must be 16-byte aligned, lddz %% 4 == 0");
    HeadTrainArgs a;
    a.h = h; a.ldh = ldh; a.w = w; a.bias = bias; a.add = add; a.ldadd = ldadd; a.y = y;
    a.logit = logit; a.dlogit = dlogit; a.dz = dz; a.lddz = lddz; a.ws = workspace;
    FX_CHECK_ARG(mask_from < 0 || mask_from % 4 == 0, "fx_head_train: mask_from must be a multiple of 4");
    a.M = M; a.K = K; a.root_scale = root_scale; a.mask_from = mask_from;
    const int64_t G = fx_head_train_groups(M);
    hipStream_t s = fx_hip_stream(stream);
    const dim3 grid((unsigned)G), block(256);
    const int64_t nu = fx_ceil_div(K, 256);
    if (nu <= 1) hipLaunchKernelGGL(k_head_train<1>, grid, block, 0, s, a);
    else if (nu <= 2) hipLaunchKernelGGL(k_head_train<2>, grid, block, 0, s, a);
    else if (nu <= 4) hipLaunchKernelGGL(k_head_train<4>, grid, block, 0, s, a);
    else hipLaunchKernelGGL(k_head_train<8>, grid, block, 0, s, a);
    FX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_head_reduce, dim3((unsigned)fx_ceil_div(K, 8)), block, 0, s, workspace, G, K,
                       1.f / (float)M, dW, db, loss);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
