// gemm_bf16x6_lab.hip — feasibility lab: an fp32-accurate GEMM on the bf16 matrix cores.
//
// Every fp32 operand is split EXACTLY into three bf16 pieces a = a1 + a2 + a3 (8 + 8 + 8 significand
// bits); a product keeps the six piece products of order <= 2^-16 relative,
//     a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1),
// dropping terms of 2^-24 and below — the size of an fp32 rounding — and accumulates in fp32 on
// v_mfma_f32_32x32x16_bf16 (16 k per instruction: the accumulator chain is K / 16 long where the fp32
// MFMA's is K / 2).  Six bf16 MFMAs per fp32 product = 2500 / 6 = 417 TFLOP/s fp32-equivalent at the dense
// bf16 peak, against 157 TFLOP/s of the fp32 MFMA (SURVEY.md section 7-6 names the split-bf16 form as the
// parity-mode alternative to fp32 inputs).
//
// C[M,N] = A[M,K] . B[N,K]^T, both operands K-contiguous ("NT": y = x W^T of nn.Linear as stored).
// Planes: bf16 [3][rows][K].  Tile 128 x 128 x 32, 4 waves (2 x 2, 64 x 64 each = 2 x 2 MFMA blocks), one
// LDS stage + register prefetch, two workgroups per CU.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 gemm_bf16x6_lab.hip -o gemm_bf16x6_lab
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ uint16_t f2bf_rne(float f) {
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float((uint32_t)h << 16); }

// fp32 [rows, K] (row stride ld) -> three bf16 planes [3][rows][K]
__global__ __launch_bounds__(256) void k_split3(const float* __restrict__ x, int64_t ld, int64_t rows,
                                                int64_t K, uint16_t* __restrict__ planes) {
    const int64_t n4 = rows * (K / 4);
    const int64_t plane = rows * K;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / (K / 4), c = (i - r * (K / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        const float a[4] = {v.x, v.y, v.z, v.w};
        uint16_t p1[4], p2[4], p3[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            p1[j] = f2bf_rne(a[j]);
            const float r1 = a[j] - bf2f(p1[j]);
            p2[j] = f2bf_rne(r1);
            const float r2 = r1 - bf2f(p2[j]);
            p3[j] = f2bf_rne(r2);
        }
        const int64_t o = r * K + c;
        *reinterpret_cast<uint2*>(planes + o) = make_uint2(p1[0] | ((uint32_t)p1[1] << 16), p1[2] | ((uint32_t)p1[3] << 16));
        *reinterpret_cast<uint2*>(planes + plane + o) = make_uint2(p2[0] | ((uint32_t)p2[1] << 16), p2[2] | ((uint32_t)p2[3] << 16));
        *reinterpret_cast<uint2*>(planes + 2 * plane + o) = make_uint2(p3[0] | ((uint32_t)p3[1] << 16), p3[2] | ((uint32_t)p3[3] << 16));
    }
}

#define BM 128
#define BN 128
#define BK 32
#define ROWB 80                  // bytes of one LDS row: 32 bf16 + 16 bytes of padding (conflict-free b128)
#define PLANE_B (128 * ROWB)     // one plane of one operand tile
#define OPER_B (3 * PLANE_B)

template <int NTERMS, bool EARLY>
__global__ __launch_bounds__(256, 2) void k_gemm_bf16x(const uint16_t* __restrict__ A3,
                                                       const uint16_t* __restrict__ B3,
                                                       float* __restrict__ C, int M, int N, int K,
                                                       int64_t ldc) {
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * OPER_B];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = N / BN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 — give every XCD a contiguous range of tiles
    // (whole tile rows), so that an A panel is read by one XCD's L2 only
    const int nwg = gridDim.x;
    const int bid = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
    // this thread's 12 chunks of a k tile: q = tid + 256 i, i < 6 (A) and the same for B
    // q -> chunk c = q & 3 (8 bf16), row = (q >> 2) & 127, plane = q >> 9
    u32x4 ra[6], rb[6];
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int nk = K / BK;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = tid + 256 * i;
        const int c = q & 3, row = (q >> 2) & 127, pl = q >> 9;
        ra[i] = *reinterpret_cast<const u32x4*>(A3 + pl * planeA + (m0 + row) * K + 0 * BK + c * 8);
        rb[i] = *reinterpret_cast<const u32x4*>(B3 + pl * planeB + (n0 + row) * K + 0 * BK + c * 8);
    }
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();                       // everyone is done reading the previous tile
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = tid + 256 * i;
            const int c = q & 3, row = (q >> 2) & 127, pl = q >> 9;
            *reinterpret_cast<u32x4*>(lds + pl * PLANE_B + row * ROWB + c * 16) = ra[i];
            *reinterpret_cast<u32x4*>(lds + OPER_B + pl * PLANE_B + row * ROWB + c * 16) = rb[i];
        }
        __syncthreads();
        const int kn = kt + 1 < nk ? kt + 1 : kt;    // (the last iteration re-loads its own tile: no branch)
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int q = tid + 256 * i;
            const int c = q & 3, row = (q >> 2) & 127, pl = q >> 9;
            ra[i] = *reinterpret_cast<const u32x4*>(A3 + pl * planeA + (m0 + row) * K + kn * BK + c * 8);
            rb[i] = *reinterpret_cast<const u32x4*>(B3 + pl * planeB + (n0 + row) * K + kn * BK + c * 8);
        }
        if (EARLY) __builtin_amdgcn_sched_barrier(0);      // keep the prefetch in front of the MFMAs
#pragma unroll
        for (int s = 0; s < 2; ++s) {          // two k16 steps per tile
            bf16x8 fa[3][2], fb[3][2];
            const int c = 2 * s + (lane >> 5);
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rowa = wm * 64 + b * 32 + (lane & 31);
                    const int rowb = wn * 64 + b * 32 + (lane & 31);
                    fa[p][b] = *reinterpret_cast<const bf16x8*>(lds + p * PLANE_B + rowa * ROWB + c * 16);
                    fb[p][b] = *reinterpret_cast<const bf16x8*>(lds + OPER_B + p * PLANE_B + rowb * ROWB + c * 16);
                }
            // operands swapped (B first): the accumulators hold the TRANSPOSED block, so that a lane owns
            // one row m of C and four consecutive columns per register group (16-byte stores)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (NTERMS >= 6) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2][j], fa[0][i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[2][i], acc[i][j], 0, 0, 0);
                    }
                    if (NTERMS >= 3) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][j], fa[0][i], acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[1][i], acc[i][j], 0, 0, 0);
                    }
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][j], fa[0][i], acc[i][j], 0, 0, 0);
                }
        }
    }
    // C: lane -> m = (lane & 31), n = 8 g + 4 (lane >> 5) + (0..3) inside the 32 x 32 block
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + wm * 64 + i * 32 + (lane & 31);
            const int64_t nb = n0 + wn * 64 + j * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 v = make_float4(acc[i][j][4 * g + 0], acc[i][j][4 * g + 1],
                                             acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                *reinterpret_cast<float4*>(C + m * ldc + nb + 8 * g) = v;
            }
        }
}

// Second structure: 8 waves (2 x 4, a wave owns 64 x 32 of the tile: two waves per SIMD, so one can issue
// MFMAs while its partner waits for LDS or the barrier), TWO LDS stages and ONE barrier per k tile:
//   compute tile t from stage t & 1 | write the registers of tile t + 1 into the other stage | request
//   tile t + 2 from memory | barrier
template <int NTERMS>
__global__ __launch_bounds__(512, 2) void k_gemm_bf16x_w8(const uint16_t* __restrict__ A3,
                                                          const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int M, int N, int K,
                                                          int64_t ldc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds2[];      // 2 stages x 2 operands
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int tiles_n = N / BN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 — give every XCD a contiguous range of tiles
    // (whole tile rows), so that an A panel is read by one XCD's L2 only
    const int nwg = gridDim.x;
    const int bid = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
    u32x4 ra[3], rb[3];
    const int nk = K / BK;
#define W8_GLOAD(KT)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const int q = tid + 512 * i;                                                                   \
        const int c = q & 3, row = (q >> 2) & 127, pl = q >> 9;                                        \
        ra[i] = *reinterpret_cast<const u32x4*>(A3 + pl * planeA + (m0 + row) * K + (KT) * BK + c * 8); \
        rb[i] = *reinterpret_cast<const u32x4*>(B3 + pl * planeB + (n0 + row) * K + (KT) * BK + c * 8); \
    }
#define W8_LSTORE(ST)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                                    \
        const int q = tid + 512 * i;                                                                   \
        const int c = q & 3, row = (q >> 2) & 127, pl = q >> 9;                                        \
        *reinterpret_cast<u32x4*>(lds2 + (ST) * 2 * OPER_B + pl * PLANE_B + row * ROWB + c * 16) = ra[i]; \
        *reinterpret_cast<u32x4*>(lds2 + (ST) * 2 * OPER_B + OPER_B + pl * PLANE_B + row * ROWB + c * 16) = rb[i]; \
    }
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    W8_GLOAD(0)
    W8_LSTORE(0)
    {
        const int k1 = nk > 1 ? 1 : 0;
        W8_GLOAD(k1)
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const unsigned char* st = lds2 + (kt & 1) * 2 * OPER_B;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[3][2], fb[3];
            const int c = 2 * s + (lane >> 5);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const int rowa = wm * 64 + b * 32 + (lane & 31);
                    fa[p][b] = *reinterpret_cast<const bf16x8*>(st + p * PLANE_B + rowa * ROWB + c * 16);
                }
                const int rowb = wn * 32 + (lane & 31);
                fb[p] = *reinterpret_cast<const bf16x8*>(st + OPER_B + p * PLANE_B + rowb * ROWB + c * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (NTERMS >= 6) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2], fa[0][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1], fa[1][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[2][i], acc[i], 0, 0, 0);
                }
                if (NTERMS >= 3) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1], fa[0][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[1][i], acc[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[0][i], acc[i], 0, 0, 0);
            }
        }
        // the other stage was last read in iteration kt - 1, before the barrier that ended it
        if (kt + 1 < nk) {
            W8_LSTORE((kt + 1) & 1)
        }
        {
            const int kn = kt + 2 < nk ? kt + 2 : kt;
            W8_GLOAD(kn)
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + i * 32 + (lane & 31);
        const int64_t nb = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2],
                                         acc[i][4 * g + 3]);
            *reinterpret_cast<float4*>(C + m * ldc + nb + 8 * g) = v;
        }
    }
}

// Third structure: LDS-DMA (global_load_lds_dwordx4) into THREE dense LDS stages, prefetch distance two,
// ONE raw s_barrier per k tile and a counted s_waitcnt vmcnt — the loads of tile t + 1 stay in flight across
// the barrier that releases tile t.  The LDS image is dense ([row][64 B], what the DMA can write: wave-uniform
// base + lane x 16 B); bank conflicts of the fragment reads are avoided by XOR-ing the 16-byte chunk index
// with (row >> 2) & 3 on BOTH sides (the per-lane GLOBAL address picks the chunk, the ds_read address undoes it).
#define G3_PLANE (128 * 64)
#define G3_OPER (3 * G3_PLANE)
#define G3_STAGE (2 * G3_OPER)
template <int NTERMS>
__global__ __launch_bounds__(512, 2) void k_gemm_bf16x_g3(const uint16_t* __restrict__ A3,
                                                          const uint16_t* __restrict__ B3,
                                                          float* __restrict__ C, int M, int N, int K,
                                                          int64_t ldc) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds3[];      // 3 stages
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int tiles_n = N / BN;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 — give every XCD a contiguous range of tiles
    // (whole tile rows), so that an A panel is read by one XCD's L2 only
    const int nwg = gridDim.x;
    const int bid = (nwg % 8 == 0) ? (int)(blockIdx.x % 8) * (nwg / 8) + (int)(blockIdx.x / 8) : (int)blockIdx.x;
    const int tm = bid / tiles_n, tn = bid % tiles_n;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;
    const int64_t planeA = (int64_t)M * K, planeB = (int64_t)N * K;
    const int nk = K / BK;
    // this wave's six DMA units of a tile: unit u = w * 6 + j -> operand u / 24, plane (u / 8) % 3, 16-row
    // block u % 8; lane l -> row (l >> 2) of the block, LDS chunk slot l & 3, global chunk slot ^ swizzle
    const uint16_t* gsrc[6];
    uint32_t ldst[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int u = w * 6 + j;
        const int oper = u / 24, pl = (u >> 3) % 3, blk = u & 7;
        const int row = blk * 16 + (lane >> 2);
        const int c = (lane & 3) ^ ((row >> 2) & 3);
        gsrc[j] = (oper ? B3 + pl * planeB + (n0 + row) * K : A3 + pl * planeA + (m0 + row) * K) + c * 8;
        ldst[j] = (uint32_t)(oper * G3_OPER + pl * G3_PLANE + blk * 1024);     // wave-uniform base
    }
    const uint32_t lds_base = (uint32_t)reinterpret_cast<uintptr_t>(lds3);
#define G3_ISSUE(KT, ST)                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 6; ++j)                                                          \
        __builtin_amdgcn_global_load_lds(                                                                \
            (const __attribute__((address_space(1))) void*)(uintptr_t)(gsrc[j] + (int64_t)(KT) * BK),   \
            (__attribute__((address_space(3))) void*)(uintptr_t)(lds_base + (uint32_t)(ST) * G3_STAGE + ldst[j]),  \
            16, 0, 0);
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    G3_ISSUE(0, 0)
    {
        const int k1 = nk > 1 ? 1 : 0;
        G3_ISSUE(k1, 1)
    }
    // fragment addresses (byte offsets inside a stage), constant over the k loop
    uint32_t offa[2][2], offb[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int c = 2 * s + (lane >> 5);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = wm * 64 + b * 32 + (lane & 31);
            offa[s][b] = (uint32_t)(r * 64 + ((c ^ ((r >> 2) & 3)) * 16));
        }
        const int rb = wn * 32 + (lane & 31);
        offb[s] = (uint32_t)(G3_OPER + rb * 64 + ((c ^ ((rb >> 2) & 3)) * 16));
    }
    for (int kt = 0; kt < nk; ++kt) {
        // tile kt has landed once at most the 6 DMAs of tile kt + 1 are still outstanding (per wave), and
        // every wave has said so
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        {
            // stage (kt + 2) % 3 was read in iteration kt - 1: every wave is past those reads (it waited for
            // them before its MFMAs) when it arrives at the barrier above
            const int kn = kt + 2 < nk ? kt + 2 : nk - 1;
            const int stn = (kt + 2) % 3;
            G3_ISSUE(kn, stn)
        }
        const unsigned char* st = lds3 + (kt % 3) * G3_STAGE;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[3][2], fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    fa[p][b] = *reinterpret_cast<const bf16x8*>(st + p * G3_PLANE + offa[s][b]);
                fb[p] = *reinterpret_cast<const bf16x8*>(st + p * G3_PLANE + offb[s]);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (NTERMS >= 6) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2], fa[0][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1], fa[1][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[2][i], acc[i], 0, 0, 0);
                }
                if (NTERMS >= 3) {
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1], fa[0][i], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[1][i], acc[i], 0, 0, 0);
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0], fa[0][i], acc[i], 0, 0, 0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + i * 32 + (lane & 31);
        const int64_t nb = n0 + wn * 32 + 4 * (lane >> 5);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 v = make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2],
                                         acc[i][4 * g + 3]);
            *reinterpret_cast<float4*>(C + m * ldc + nb + 8 * g) = v;
        }
    }
}

template <int NT>
static float run_g3(const uint16_t* A3, const uint16_t* B3, float* C, int M, int N, int K, int reps) {
    dim3 grid((M / BM) * (N / BN));
    const size_t smem = 3 * G3_STAGE;
    HC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16x_g3<NT>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((k_gemm_bf16x_g3<NT>), grid, dim3(512), smem, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_gemm_bf16x_g3<NT>), grid, dim3(512), smem, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipEventRecord(e1, 0));
    HC(hipDeviceSynchronize());
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

template <int NT>
static float run_w8(const uint16_t* A3, const uint16_t* B3, float* C, int M, int N, int K, int reps) {
    dim3 grid((M / BM) * (N / BN));
    const size_t smem = 4 * OPER_B;
    HC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_bf16x_w8<NT>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    hipLaunchKernelGGL((k_gemm_bf16x_w8<NT>), grid, dim3(512), smem, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_gemm_bf16x_w8<NT>), grid, dim3(512), smem, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipEventRecord(e1, 0));
    HC(hipDeviceSynchronize());
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

// references on a sample of rows: fp64 sums and the k-ordered fp32 fmaf chain (what the fp32 MFMA computes)
__global__ void k_ref(const float* A, const float* B, int M, int N, int K, int row_step, double* C64,
                      float* C32) {
    const int64_t rows = (M + row_step - 1) / row_step;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * N; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N, n = i - r * N, m = r * row_step;
        double s = 0.0;
        float f = 0.f;
        for (int k = 0; k < K; ++k) {
            s += (double)A[m * K + k] * (double)B[n * K + k];
            f = fmaf(A[m * K + k], B[n * K + k], f);
        }
        C64[i] = s;
        C32[i] = f;
    }
}

__global__ void k_fill(float* p, int64_t n, uint32_t seed, float scale) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale;
    }
}

template <int NT, bool EARLY>
static float run(const uint16_t* A3, const uint16_t* B3, float* C, int M, int N, int K, int reps) {
    dim3 grid((M / BM) * (N / BN));
    hipLaunchKernelGGL((k_gemm_bf16x<NT, EARLY>), grid, dim3(256), 0, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r)
        hipLaunchKernelGGL((k_gemm_bf16x<NT, EARLY>), grid, dim3(256), 0, 0, A3, B3, C, M, N, K, (int64_t)N);
    HC(hipEventRecord(e1, 0));
    HC(hipDeviceSynchronize());
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

static void bench(int M, int N, int K, const char* what) {
    float *A, *B, *C;
    uint16_t *A3, *B3;
    HC(hipMalloc(&A, (size_t)M * K * 4)); HC(hipMalloc(&B, (size_t)N * K * 4)); HC(hipMalloc(&C, (size_t)M * N * 4));
    HC(hipMalloc(&A3, (size_t)3 * M * K * 2)); HC(hipMalloc(&B3, (size_t)3 * N * K * 2));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 17u, 2.f);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, B, (int64_t)N * K, 91u, 0.1f);
    // split timings
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_split3, dim3(4096), dim3(256), 0, 0, A, (int64_t)K, (int64_t)M, (int64_t)K, A3);
    hipLaunchKernelGGL(k_split3, dim3(4096), dim3(256), 0, 0, B, (int64_t)K, (int64_t)N, (int64_t)K, B3);
    HC(hipDeviceSynchronize());
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < 10; ++r)
        hipLaunchKernelGGL(k_split3, dim3(4096), dim3(256), 0, 0, A, (int64_t)K, (int64_t)M, (int64_t)K, A3);
    HC(hipEventRecord(e1, 0));
    HC(hipDeviceSynchronize());
    float ms_split; HC(hipEventElapsedTime(&ms_split, e0, e1));
    const double flop = 2.0 * M * N * K;
    const float t6 = run<6, true>(A3, B3, C, M, N, K, 20);
    // accuracy of the 6-term result on every 37th row
    const int step = 37;
    const int64_t rows = (M + step - 1) / step;
    double* C64; float* C32;
    HC(hipMalloc(&C64, (size_t)rows * N * 8)); HC(hipMalloc(&C32, (size_t)rows * N * 4));
    hipLaunchKernelGGL(k_ref, dim3(1024), dim3(256), 0, 0, A, B, M, N, K, step, C64, C32);
    std::vector<double> h64(rows * N);
    std::vector<float> h32(rows * N), hc((size_t)M * N);
    HC(hipMemcpy(h64.data(), C64, rows * N * 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(h32.data(), C32, rows * N * 4, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hc.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    double e6 = 0, e32 = 0, nrm = 0, m6 = 0, m32 = 0, mx = 0;
    for (int64_t r = 0; r < rows; ++r)
        for (int n = 0; n < N; ++n) {
            const double ref = h64[r * N + n];
            const double d6 = hc[(size_t)(r * step) * N + n] - ref, d32 = h32[r * N + n] - ref;
            e6 += d6 * d6; e32 += d32 * d32; nrm += ref * ref;
            if (fabs(d6) > m6) m6 = fabs(d6);
            if (fabs(d32) > m32) m32 = fabs(d32);
            if (fabs(ref) > mx) mx = fabs(ref);
        }
    const float t3 = run<3, true>(A3, B3, C, M, N, K, 20);
    const float t6l = run<6, false>(A3, B3, C, M, N, K, 20);
    const float t1 = run<1, true>(A3, B3, C, M, N, K, 20);
    const float w6 = run_w8<6>(A3, B3, C, M, N, K, 20);
    {   // the 8-wave kernel's result against the 4-wave one's (same products, same k order per element)
        std::vector<float> hw((size_t)M * N);
        HC(hipMemcpy(hw.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < hw.size(); ++i) bad += hw[i] != hc[i];
        printf("%-34s 8 waves, 2 LDS stages: x6 %7.2f us %6.1f TF-eq, x1 %7.2f us; %zu elements differ from the 4-wave result\n",
               "", w6, flop / w6 * 1e-6, run_w8<1>(A3, B3, C, M, N, K, 20), bad);
    }
    {
        const float g6 = run_g3<6>(A3, B3, C, M, N, K, 20);
        std::vector<float> hw((size_t)M * N);
        HC(hipMemcpy(hw.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < hw.size(); ++i) bad += hw[i] != hc[i];
        printf("%-34s LDS-DMA, 3 stages, 1 raw barrier: x6 %7.2f us %6.1f TF-eq, x1 %7.2f us; %zu elements differ from the 4-wave result\n",
               "", g6, flop / g6 * 1e-6, run_g3<1>(A3, B3, C, M, N, K, 20), bad);
    }
    printf("%-34s %4dx%4dx%4d  x6 %7.2f us %6.1f TF-eq (loads sunk by the compiler: %7.2f us) | x3 %7.2f us | x1 (plain bf16) %7.2f us %6.1f TF | split A %6.2f us\n",
           what, M, N, K, t6, flop / t6 * 1e-6, t6l, t3, t1, flop / t1 * 1e-6, 1e3f * ms_split / 10);
    printf("%-34s rel L2 error vs fp64: x6 %.3e   fp32 fma chain %.3e   | max |err| / max |c|: x6 %.3e   chain %.3e\n",
           "", sqrt(e6 / nrm), sqrt(e32 / nrm), m6 / mx, m32 / mx);
    hipFree(A); hipFree(B); hipFree(C); hipFree(A3); hipFree(B3); hipFree(C64); hipFree(C32);
}

int main() {
    bench(4096, 1024, 1024, "tower fwd / dX 1024-wide");
    bench(4096, 1024, 640, "tower fwd, first layer (K 640)");
    bench(4096, 640, 1024, "dX of the first layer");
    bench(1024, 1024, 4096, "dW 1024 x 1024 (K = batch)");
    bench(1024, 640, 4096, "dW first layer");
    bench(4096, 4096, 4096, "4096^3");
    return 0;
}
