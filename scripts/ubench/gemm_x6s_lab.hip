// gemm_x6s_lab.hip — round 5 lab: fp32-accurate GEMM on the bf16 matrix cores with the operand split done
// INSIDE the kernel (fp32 in memory, three exact bf16 planes only ever exist in LDS).
//
// Round 4's lab (gemm_bf16x6_lab.hip) read pre-split planes: 6 bytes per element through L2 -> LDS plus a
// separate split launch per operand, and stopped at 55 us on 4096 x 1024 x 1024 (fp32 MFMA kernel: 71-76).
// Here a workgroup loads fp32 (4 bytes per element, no extra launch, no plane tensors in HBM), each thread
// splits the 16 elements it staged (v_cvt_pk_bf16_f32 + exact residuals) and writes the planes to LDS;
// the six piece products per fp32 product run on v_mfma_f32_32x32x16_bf16 as before.
//
// Tile 128 x 128 x 32, 8 waves (2 x 4: a wave owns 64 x 32 = two 32x32 accumulators), two LDS stages of three
// planes per operand ([plane][row][80 B]: 32 bf16 + 16 B pad, conflict-free ds_read_b128), ONE barrier per
// k tile in the middle of it.  Per k tile and wave: 24 MFMAs in 24 "slots"; behind every MFMA a fixed share
// of the tile's other work is issued (global loads of tile t+2, split of tile t+1, plane writes, fragment
// reads), pinned with sched_barrier.
//   first half  (k16 step 0): loads of tile t+2 | B planes of t+1 (split in the previous half) -> LDS |
//                             split A of t+1 -> LDS | fragment reads of step 1     | barrier
//   second half (k16 step 1): split B of t+2 (registers) | fragment reads of step 0 of tile t+1
// Operand layouts: KC (k contiguous: x W^T forward) or row contiguous (dX's W, both operands of dW) — a
// row-contiguous operand is loaded as 8 coalesced dwords (one row, 8 k) per thread, so both kinds produce
// the same LDS image and no transposing read is needed.
//
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off gemm_x6s_lab.hip -o gemm_x6s_lab -ldl
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <type_traits>
#include <vector>

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define X6_ROWB 80
#define X6_PLANE (128 * X6_ROWB)
#define X6_OPER (3 * X6_PLANE)
#define X6_STAGE (2 * X6_OPER)
#define X6_LDS (2 * X6_STAGE)

struct X6Args {
    const float* A; int64_t lda;
    const float* B; int64_t ldb;
    float* C; int64_t ldc;
    int M, N, K, k_chunk;
    const float* bias; int relu;
    float* ws; int split_k;
};

template <int I, int N, typename F>
__device__ __forceinline__ void sfor(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        sfor<I + 1, N>(f);
    }
}

__device__ __forceinline__ uint32_t pk_bf16(float a, float b) {
    f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

#ifdef X6_ASM_SUB
__device__ __forceinline__ float xsub(float a, float b) {
    float r;
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#else
__device__ __forceinline__ float xsub(float a, float b) { return a - b; }
#endif

// one operand tile's staging: 8 floats per thread, their three planes as 4 packed dwords each
template <bool KC>
struct Opnd {
    static constexpr int NL = KC ? 2 : 8;     // global load instructions per tile
    static constexpr int NW = KC ? 6 : 3;     // LDS write instructions per tile
    const char* base;
    int64_t tstride, kstride;
    uint32_t voff[2];
    uint32_t woff[2];

    __device__ __forceinline__ void init(const float* P, int64_t ld, int64_t r0, int64_t kbeg) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
            base = reinterpret_cast<const char*>(P + r0 * ld + kbeg);
            tstride = 32 * 4;
            kstride = 0;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int q = tid + 512 * p;
                // rows of a wave permuted (bits 0 and 2 swapped): the 16 lanes one ds_write_b64 group serves
                // hold rows R and R + 4 (80 dwords apart = 16 mod 32 banks) instead of R and R + 1 (20 apart:
                // banks 0-3 hit twice — every plane write ran 2-way conflicted, SQ_LDS_BANK_CONFLICT)
                const int rl = q >> 3;
                const int row = (rl & ~5) | ((rl & 1) << 2) | ((rl >> 2) & 1), k4 = (q & 7) * 4;
                voff[p] = (uint32_t)((row * ld + k4) * 4);
                woff[p] = (uint32_t)(row * X6_ROWB + (q & 7) * 8);
            }
        } else {
            base = reinterpret_cast<const char*>(P + kbeg * ld + r0);
            tstride = 32 * ld * 4;
            kstride = ld * 4;
            const int row = tid & 127, o = tid >> 7;
            voff[0] = (uint32_t)((8 * o * ld + row) * 4);
            voff[1] = 0;
            woff[0] = (uint32_t)(row * X6_ROWB + o * 16);
            woff[1] = 0;
        }
    }
    template <int I>
    __device__ __forceinline__ void load_piece(int64_t t, float (&r)[8]) const {
#ifdef X6_NO_GLOAD
        if (t > 1) return;
#endif
        const char* b = base + t * tstride;
        if constexpr (KC) {
            const float4 v = *reinterpret_cast<const float4*>(b + voff[I]);
            r[4 * I + 0] = v.x; r[4 * I + 1] = v.y; r[4 * I + 2] = v.z; r[4 * I + 3] = v.w;
        } else {
            r[I] = *reinterpret_cast<const float*>(b + I * kstride + voff[0]);
        }
    }
    // split sub-op S (0..11): pair j = S / 3, plane step = S % 3; r is overwritten by the residuals
    template <int S>
    static __device__ __forceinline__ void split_piece(float (&r)[8], uint32_t (&pl)[3][4]) {
        constexpr int j = S / 3, st = S % 3;
        asm volatile("" : "+v"(r[2 * j]), "+v"(r[2 * j + 1]));      // anchor: not before this slot
#ifdef X6_NO_SPLIT
        pl[st][j] = __float_as_uint(r[2 * j]) ^ (st * 77u);
        return;
#endif
        const uint32_t p = pk_bf16(r[2 * j], r[2 * j + 1]);
        pl[st][j] = p;
        if constexpr (st < 2) {
            r[2 * j] = xsub(r[2 * j], __uint_as_float(p << 16));
            r[2 * j + 1] = xsub(r[2 * j + 1], __uint_as_float(p & 0xffff0000u));
            asm volatile("" : "+v"(r[2 * j]), "+v"(r[2 * j + 1]));  // anchor: not after this slot
        } else {
            asm volatile("" : "+v"(pl[st][j]));
        }
    }
    // LDS write W (0..NW-1) of the planes into operand image `dst` (byte pointer to plane 0)
    template <int W>
    __device__ __forceinline__ void write_piece(unsigned char* dst, const uint32_t (&pl)[3][4]) const {
#ifdef X6_NO_LDSW
        if (pl[0][0] == 0x12345u) return;     // (practically never true: keeps the planes live)
        else if (pl[1][1] != 0x54321u) return;
#endif
        if constexpr (KC) {
            constexpr int g = W / 3, p = W % 3;     // float4 g (rows +64 g), plane p
            u32x2 v = {pl[p][2 * g], pl[p][2 * g + 1]};
            *reinterpret_cast<u32x2*>(dst + p * X6_PLANE + woff[g]) = v;
        } else {
            u32x4 v = {pl[W][0], pl[W][1], pl[W][2], pl[W][3]};
            *reinterpret_cast<u32x4*>(dst + W * X6_PLANE + woff[0]) = v;
        }
    }
};

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_x6s(X6Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    using OA = Opnd<A_KC>;
    using OB = Opnd<B_KC>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, half = lane >> 5;
    const int tiles_n = a.N / 128, tiles_m = a.M / 128;
    const int nwg = tiles_m * tiles_n;
    int T = blockIdx.x;
    if (nwg >= 8) {
        const int q = nwg >> 3, r = nwg & 7, xcd = T & 7;
        T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (T >> 3);
    }
    const int64_t m0 = (int64_t)(T / tiles_n) * 128, n0 = (int64_t)(T % tiles_n) * 128;
    const int z = blockIdx.y;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = kbeg + a.k_chunk < a.K ? kbeg + a.k_chunk : a.K;
    const int nk = (int)((kend - kbeg) / 32);

    OA oa;
    OB ob;
    oa.init(a.A, a.lda, m0, kbeg);
    ob.init(a.B, a.ldb, n0, kbeg);

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

    float ra[2][8], rb[8];
    uint32_t pa[3][4], pb[3][4];
    bf16x8 fa[2][3][2], fb[2][3];       // [k16 step][plane][block]

    const uint32_t fo_a = (uint32_t)((wm * 64 + l31) * X6_ROWB + half * 16);
    const uint32_t fo_b = (uint32_t)(X6_OPER + (wn * 32 + l31) * X6_ROWB + half * 16);

    // fragment read R (0..8) of k16 step S from stage st
    auto frag_read = [&](auto rr, auto ss, const unsigned char* st) {
        constexpr int R = decltype(rr)::value, S = decltype(ss)::value;
        // order of first use: b2 a0 | b1 a1 | b0 a2
        if constexpr (R == 0) fb[S][2] = *reinterpret_cast<const bf16x8*>(st + fo_b + 2 * X6_PLANE + S * 32);
        else if constexpr (R == 1) fa[S][0][0] = *reinterpret_cast<const bf16x8*>(st + fo_a + 0 * X6_PLANE + S * 32);
        else if constexpr (R == 2) fa[S][0][1] = *reinterpret_cast<const bf16x8*>(st + fo_a + 0 * X6_PLANE + 32 * X6_ROWB + S * 32);
        else if constexpr (R == 3) fb[S][1] = *reinterpret_cast<const bf16x8*>(st + fo_b + 1 * X6_PLANE + S * 32);
        else if constexpr (R == 4) fa[S][1][0] = *reinterpret_cast<const bf16x8*>(st + fo_a + 1 * X6_PLANE + S * 32);
        else if constexpr (R == 5) fa[S][1][1] = *reinterpret_cast<const bf16x8*>(st + fo_a + 1 * X6_PLANE + 32 * X6_ROWB + S * 32);
        else if constexpr (R == 6) fb[S][0] = *reinterpret_cast<const bf16x8*>(st + fo_b + 0 * X6_PLANE + S * 32);
        else if constexpr (R == 7) fa[S][2][0] = *reinterpret_cast<const bf16x8*>(st + fo_a + 2 * X6_PLANE + S * 32);
        else fa[S][2][1] = *reinterpret_cast<const bf16x8*>(st + fo_a + 2 * X6_PLANE + 32 * X6_ROWB + S * 32);
    };
    // MFMA m (0..11) of k16 step S: product m / 2 (order b2a0 b1a1 b0a2 b1a0 b0a1 b0a0), block m % 2
    // (the MFMA builtin is a pure operation: left alone, the instruction selector floats all 12 MFMAs of a
    // half to its top — their operands were read half a tile earlier.  The empty asm "touches" one operand
    // in the slot the MFMA belongs to; asm volatile statements and sched_barrier keep their order.)
    auto mfma = [&](auto mm, auto ss) {
        constexpr int m = decltype(mm)::value, S = decltype(ss)::value;
        constexpr int p = m / 2, i = m % 2;
        constexpr int bp = p == 0 ? 2 : (p == 1 || p == 3) ? 1 : 0;
        constexpr int ap = p == 0 ? 0 : p == 1 ? 1 : p == 2 ? 2 : p == 3 ? 0 : p == 4 ? 1 : 0;
        asm volatile("" : "+v"(fa[S][ap][i]));
#ifndef X6_NO_MFMA
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][bp], fa[S][ap][i], acc[i], 0, 0, 0);
#else
        acc[i][m] += __builtin_bit_cast(float, (uint32_t)fb[S][bp][0]) + __builtin_bit_cast(float, (uint32_t)fa[S][ap][i][0]);
#endif
    };

    // one k tile.  kt = -1 (DO = false): the pipeline's fill — everything but the MFMAs and step-1 reads.
    auto body = [&](int kt, auto par, auto domf) {
        constexpr int P = decltype(par)::value;          // kt & 1
        constexpr bool DO = decltype(domf)::value;
        unsigned char* const st_cur = lds + P * X6_STAGE;            // tile kt
        unsigned char* const st_nxt = lds + (P ^ 1) * X6_STAGE;      // tile kt + 1
        const int64_t tl = kt + 2 < nk ? kt + 2 : nk - 1;
        // ---- first half ----
        constexpr int NLB = OB::NL, NLA = OA::NL, NWB = OB::NW, NWA = OA::NW;
        // list X: [B loads][A loads][B plane writes][A split 0..5][A writes first group][A split 6..11][A writes rest]
        constexpr int NWA1 = A_KC ? 3 : 0;
        constexpr int X0 = 0, X1 = X0 + NLB, X2 = X1 + NLA, X3 = X2 + NWB, X4 = X3 + 6, X5 = X4 + NWA1,
                      X6 = X5 + 6, NX = X6 + (NWA - NWA1);
        auto xop = [&](auto ii) {
            constexpr int I = decltype(ii)::value;
            if constexpr (I < X1) ob.template load_piece<I - X0>(tl, rb);
            else if constexpr (I < X2) oa.template load_piece<I - X1>(tl, ra[P]);
            else if constexpr (I < X3) ob.template write_piece<I - X2>(st_nxt + X6_OPER, pb);
            else if constexpr (I < X4) OA::template split_piece<I - X3>(ra[P ^ 1], pa);
            else if constexpr (I < X5) {
                // KC: float4 0 (pairs 0, 1) is complete: its three planes
                constexpr int W = I - X4;                         // plane
                oa.template write_piece<W>(st_nxt, pa);           // g = 0
            } else if constexpr (I < X6) OA::template split_piece<6 + I - X5>(ra[P ^ 1], pa);
            else {
                constexpr int W = I - X6 + NWA1;
                oa.template write_piece<W>(st_nxt, pa);
            }
        };
        sfor<0, 12>([&](auto mm) {
            constexpr int m = decltype(mm)::value;
            if constexpr (DO) mfma(mm, std::integral_constant<int, 0>{});
            sfor<(m * NX) / 12, ((m + 1) * NX) / 12>(xop);
            if constexpr (DO && m < 9) frag_read(mm, std::integral_constant<int, 1>{}, st_cur);
            __builtin_amdgcn_sched_barrier(0);
        });
        // ---- second half: the barrier sits two slots in — the plane writes and step-1 reads of the first
        // half have had those slots to land, so the lgkmcnt(0) in front of it is (nearly) free ----
        sfor<0, 12>([&](auto mm) {
            constexpr int m = decltype(mm)::value;
            if constexpr (m == 2) {
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (DO) mfma(mm, std::integral_constant<int, 1>{});
            OB::template split_piece<m>(rb, pb);
            if constexpr (m >= 2 && m < 11) frag_read(std::integral_constant<int, m - 2>{}, std::integral_constant<int, 0>{}, st_nxt);
            __builtin_amdgcn_sched_barrier(0);
        });
    };

    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    if (nk > 0) {
        // fill: tile 0 raw -> rb / ra[0]; B planes of tile 0; then the MFMA-less pass for kt = -1
        sfor<0, OB::NL>([&](auto ii) { ob.template load_piece<decltype(ii)::value>(0, rb); });
        sfor<0, OA::NL>([&](auto ii) { oa.template load_piece<decltype(ii)::value>(0, ra[0]); });
        sfor<0, 12>([&](auto ss) { OB::template split_piece<decltype(ss)::value>(rb, pb); });
        body(-1, P1{}, std::false_type{});
        int kt = 0;
        for (; kt + 1 < nk; kt += 2) {
            body(kt, P0{}, std::true_type{});
            body(kt + 1, P1{}, std::true_type{});
        }
        if (kt < nk) body(kt, P0{}, std::true_type{});
    }

    // epilogue: operands were swapped (B first), so a lane owns row m = l31 of a block and, per register
    // group g, the columns 8 g + 4 half .. + 3
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int64_t m = m0 + wm * 64 + i * 32 + l31;
        const int64_t nb = n0 + wn * 32 + 4 * half;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 v = make_float4(acc[i][4 * g + 0], acc[i][4 * g + 1], acc[i][4 * g + 2], acc[i][4 * g + 3]);
            const int64_t n = nb + 8 * g;
            if (a.split_k > 1) {
                *reinterpret_cast<float4*>(a.ws + ((int64_t)z * a.M + m) * a.N + n) = v;
            } else {
                if (a.bias) {
                    const float4 bv = *reinterpret_cast<const float4*>(a.bias + n);
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                }
                if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                *reinterpret_cast<float4*>(a.C + m * a.ldc + n) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// references
__global__ void k_ref(const float* A, int64_t sa_m, int64_t sa_k, const float* B, int64_t sb_n, int64_t sb_k,
                      int M, int N, int K, int row_step, double* C64) {
    const int64_t rows = (M + row_step - 1) / row_step;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < rows * N; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / N, n = i - r * N, m = r * row_step;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += (double)A[m * sa_m + k * sa_k] * (double)B[n * sb_n + k * sb_k];
        C64[i] = s;
    }
}

static float g_fill_shift = 0.5f;      // 0.5: zero-mean data; 0: all positive (accumulators grow monotonically)
__global__ void k_fill(float* p, int64_t n, uint32_t seed, float scale, float shift) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x >> 8) * (1.0f / 16777216.0f) - shift) * scale;
    }
}

typedef int (*fx_gemm_f32_t)(int32_t, int32_t, int64_t, int64_t, int64_t, const float*, int64_t, const float*,
                             int64_t, float*, int64_t, const void*, int32_t, float*, void*);
struct fx_epi { const float* bias; float* zout; int64_t ldz; int32_t act; const float* mul; int64_t ldmul;
                const float* mask; int64_t ldmask; const float* add; int64_t ldadd; float* rowsum; };
static fx_gemm_f32_t g_fx_gemm = nullptr;

template <bool AK, bool BK_>
static float run_x6(const X6Args& a, int reps) {
    dim3 grid((a.M / 128) * (a.N / 128), a.split_k);
    HC(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_x6s<AK, BK_>),
                           hipFuncAttributeMaxDynamicSharedMemorySize, X6_LDS));
    hipLaunchKernelGGL((k_gemm_x6s<AK, BK_>), grid, dim3(512), X6_LDS, 0, a);
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
    HC(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_gemm_x6s<AK, BK_>), grid, dim3(512), X6_LDS, 0, a);
    HC(hipEventRecord(e1, 0));
    HC(hipDeviceSynchronize());
    float ms; HC(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps;
}

// ta: A stored [K, M] (row contiguous), tb: B stored [N, K] (k contiguous) — fx_gemm_f32's convention
static void bench(int M, int N, int K, int ta, int tb, int split_k, int epi, const char* what) {
    float *A, *B, *C, *ws = nullptr, *bias, *C2;
    HC(hipMalloc(&A, (size_t)M * K * 4)); HC(hipMalloc(&B, (size_t)N * K * 4));
    HC(hipMalloc(&C, (size_t)M * N * 4)); HC(hipMalloc(&C2, (size_t)M * N * 4)); HC(hipMalloc(&bias, (size_t)N * 4));
    HC(hipMalloc(&ws, (size_t)(split_k > 1 ? split_k : 1) * M * (N + 1) * 4));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, A, (int64_t)M * K, 17u, 2.f, g_fill_shift);
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, B, (int64_t)N * K, 91u, 0.1f, g_fill_shift);
    hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, bias, (int64_t)N, 5u, 0.5f, 0.5f);
    X6Args a;
    memset(&a, 0, sizeof(a));
    a.A = A; a.lda = ta ? M : K; a.B = B; a.ldb = tb ? K : N; a.C = C; a.ldc = N;
    a.M = M; a.N = N; a.K = K; a.split_k = split_k; a.k_chunk = K / split_k; a.ws = ws;
    a.bias = epi ? bias : nullptr; a.relu = epi;
    const int reps = 20;
    float t;
    if (!ta && tb) t = run_x6<true, true>(a, reps);
    else if (!ta && !tb) t = run_x6<true, false>(a, reps);
    else if (ta && !tb) t = run_x6<false, false>(a, reps);
    else t = run_x6<false, true>(a, reps);
    // accuracy vs fp64 on every 37th row (split_k > 1: slabs summed on the host)
    const int step = 37;
    const int64_t rows = (M + step - 1) / step;
    double* C64; HC(hipMalloc(&C64, (size_t)rows * N * 8));
    hipLaunchKernelGGL(k_ref, dim3(1024), dim3(256), 0, 0, A, ta ? (int64_t)1 : (int64_t)K, ta ? (int64_t)M : (int64_t)1,
                       B, tb ? (int64_t)K : (int64_t)1, tb ? (int64_t)1 : (int64_t)N, M, N, K, step, C64);
    std::vector<double> h64(rows * N);
    std::vector<float> hc((size_t)M * N), hb(N);
    HC(hipMemcpy(h64.data(), C64, rows * N * 8, hipMemcpyDeviceToHost));
    HC(hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost));
    if (split_k > 1) {
        std::vector<float> hw((size_t)split_k * M * N);
        HC(hipMemcpy(hw.data(), ws, hw.size() * 4, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < (size_t)M * N; ++i) {
            float s = 0.f;
            for (int zz = 0; zz < split_k; ++zz) s += hw[(size_t)zz * M * N + i];
            hc[i] = s;
        }
    } else {
        HC(hipMemcpy(hc.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
    }
    double e6 = 0, nrm = 0, m6 = 0, mx = 0, sgn6 = 0, sgnf = 0, sabs = 0;
    for (int64_t r = 0; r < rows; ++r)
        for (int n = 0; n < N; ++n) {
            double ref = h64[r * N + n];
            if (epi && split_k == 1) { ref += hb[n]; if (ref < 0) ref = 0; }
            const double d6 = hc[(size_t)(r * step) * N + n] - ref;
            sgn6 += d6 * (ref >= 0 ? 1.0 : -1.0); sabs += fabs(ref);
            e6 += d6 * d6; nrm += ref * ref;
            if (fabs(d6) > m6) m6 = fabs(d6);
            if (fabs(ref) > mx) mx = fabs(ref);
        }
    // the production fp32-MFMA kernel on the same operands (same box, same process)
    float tf = -1.f;
    double ef = 0, mf = 0;
    if (g_fx_gemm) {
        fx_epi e; memset(&e, 0, sizeof(e));
        if (epi) { e.bias = bias; e.act = 1; }
        for (int rep = 0; rep < 2; ++rep) g_fx_gemm(ta, tb, M, N, K, A, a.lda, B, a.ldb, C2, N, &e, split_k, ws, nullptr);
        HC(hipDeviceSynchronize());
        hipEvent_t e0, e1; HC(hipEventCreate(&e0)); HC(hipEventCreate(&e1));
        HC(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) g_fx_gemm(ta, tb, M, N, K, A, a.lda, B, a.ldb, C2, N, &e, split_k, ws, nullptr);
        HC(hipEventRecord(e1, 0));
        HC(hipDeviceSynchronize());
        float ms; HC(hipEventElapsedTime(&ms, e0, e1));
        tf = 1e3f * ms / reps;
        std::vector<float> hf((size_t)M * N);
        HC(hipMemcpy(hf.data(), C2, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        if (!epi || split_k == 1) {       // every element against the fp32 kernel's (races would show as tiles)
            double dmax = 0, cmax = 0;
            size_t big = 0;
            for (size_t i = 0; i < hf.size(); ++i) {
                const double d = fabs((double)hf[i] - (double)hc[i]);
                if (d > dmax) dmax = d;
                if (fabs(hf[i]) > cmax) cmax = fabs(hf[i]);
            }
            for (size_t i = 0; i < hf.size(); ++i) big += fabs((double)hf[i] - (double)hc[i]) > 2e-5 * cmax;
            printf("    all %zu elements vs fp32-mfma: max |d| / max |c| = %.3e, %zu beyond 2e-5\n", hf.size(), dmax / cmax, big);
        }
        for (int64_t r = 0; r < rows; ++r)
            for (int n = 0; n < N; ++n) {
                double ref = h64[r * N + n];
                if (epi) { ref += hb[n]; if (ref < 0) ref = 0; }
                const double d = hf[(size_t)(r * step) * N + n] - ref;
                sgnf += d * (ref >= 0 ? 1.0 : -1.0);
                ef += d * d;
                if (fabs(d) > mf) mf = fabs(d);
            }
    }
    const double flop = 2.0 * M * N * K;
    printf("%-30s %4dx%4dx%4d ta%d tb%d sk%d epi%d | x6s %7.2f us %6.1f TF-eq | fp32-mfma %7.2f us %6.1f TF%s | relL2 x6s %.3e fp32 %.3e | max/max x6s %.3e fp32 %.3e\n",
           what, M, N, K, ta, tb, split_k, epi, t, flop / t * 1e-6, tf, tf > 0 ? flop / tf * 1e-6 : 0.0,
           split_k > 1 ? " (+reduce)" : "", sqrt(e6 / nrm), sqrt(ef / nrm), m6 / mx, mf / mx);
    printf("    signed error toward larger |c| (sum d sign(c) / sum |c|): x6s %+.3e   fx_gemm_f32 %+.3e\n", sgn6 / sabs, sgnf / sabs);
    fflush(stdout);
    hipFree(A); hipFree(B); hipFree(C); hipFree(C2); hipFree(bias); hipFree(ws); hipFree(C64);
}

int main(int argc, char** argv) {
    const char* so = argc > 1 ? argv[1] : "fuxictr_amd/libfxctr.so";
    void* h = dlopen(so, RTLD_NOW | RTLD_LOCAL);
    if (h) g_fx_gemm = (fx_gemm_f32_t)dlsym(h, "fx_gemm_f32");
    if (!g_fx_gemm) fprintf(stderr, "no libfxctr (%s): fp32-mfma column skipped\n", dlerror());
    if (argc > 2 && !strcmp(argv[2], "thr")) {      // where do 128x128-tile launches stop paying? (fx_gemm_x6_ok's threshold)
        bench(4096, 512, 1024, 0, 1, 1, 1, "fwd 128 tiles K1024");
        bench(4096, 512, 384, 0, 1, 1, 1, "fwd 128 tiles K384");
        bench(4096, 384, 1024, 0, 1, 1, 1, "fwd 96 tiles K1024");
        bench(4096, 256, 1024, 0, 1, 1, 1, "fwd 64 tiles K1024");
        bench(4096, 512, 1024, 0, 0, 1, 0, "dX 128 tiles");
        bench(2048, 1024, 1024, 0, 1, 1, 1, "fwd 128 tiles (M 2048)");
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "bias")) {
        for (int pass = 0; pass < 2; ++pass) {
            g_fill_shift = pass ? 0.f : 0.5f;
            printf("== data %s\n", pass ? "all positive" : "zero mean");
            bench(4096, 1024, 1024, 0, 1, 1, 0, "fwd 1024");
            bench(4096, 1024, 1024, 0, 0, 1, 0, "dX 1024");
            bench(1024, 1024, 4096, 1, 0, 4, 0, "dW 1024 sk4");
            bench(4096, 1024, 128, 0, 1, 1, 0, "K 128");
        }
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "prof")) {
        bench(4096, 4096, 4096, 0, 1, 1, 0, "4096^3");
        return 0;
    }
    if (argc > 2 && !strcmp(argv[2], "quick")) {
        bench(4096, 1024, 1024, 0, 1, 1, 0, "fwd 1024 (plain)");
        bench(4096, 1024, 1024, 0, 0, 1, 0, "dX 1024");
        bench(1024, 1024, 4096, 1, 0, 4, 0, "dW 1024 sk4");
        bench(4096, 4096, 4096, 0, 1, 1, 0, "4096^3");
        return 0;
    }
    bench(256, 256, 128, 0, 1, 1, 0, "small KC/KC");
    bench(256, 256, 128, 0, 0, 1, 0, "small KC/row");
    bench(256, 256, 128, 1, 0, 1, 0, "small row/row");
    bench(256, 256, 128, 1, 0, 2, 0, "small row/row splitK 2");
    bench(256, 256, 96, 0, 1, 1, 1, "small KC/KC odd nk, epi");
    bench(4096, 1024, 1024, 0, 1, 1, 1, "fwd 1024 (bias+relu)");
    bench(4096, 1024, 1024, 0, 1, 1, 0, "fwd 1024 (plain)");
    bench(4096, 1024, 640, 0, 1, 1, 1, "fwd first layer K640");
    bench(4096, 1024, 1024, 0, 0, 1, 0, "dX 1024");
    bench(4096, 640, 1024, 0, 0, 1, 0, "dX first layer");
    bench(1024, 1024, 4096, 1, 0, 4, 0, "dW 1024 sk4");
    bench(1024, 1024, 4096, 1, 0, 8, 0, "dW 1024 sk8");
    bench(1024, 640, 4096, 1, 0, 4, 0, "dW first layer sk4");
    bench(640, 640, 4096, 1, 0, 8, 0, "dW cross sk8");
    bench(4096, 640, 640, 0, 1, 1, 0, "cross fwd 640");
    bench(4096, 4096, 4096, 0, 1, 1, 0, "4096^3");
    return 0;
}
