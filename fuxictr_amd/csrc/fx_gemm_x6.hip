// fx_gemm_x6.hip — fp32-accurate GEMM on the CDNA4 bf16 matrix cores ("bf16 x 6"), operands split INSIDE the
// kernel.  Same products as fx_gemm.hip's fp32-MFMA kernels:
//   fuxictr/pytorch/layers/blocks/mlp_block.py:96            (Linear -> ReLU stack: x W^T + b)
//   fuxictr/pytorch/layers/interactions/cross_net.py:126-129 (X_{i+1} = X_i + X_0 * (W X_i + b))
// and their autograd (dX = dZ W, dW = dZ^T X, db = colsum dZ) triggered at rank_model.py:320.
//
// Arithmetic.  Every fp32 operand is split EXACTLY into three bf16 pieces a = a0 + a1 + a2 (round to nearest
// bf16, then the exact residual, twice: 8 + 8 + 8 significand bits); a product keeps the six piece products of
// order <= 2^-16 relative,
//     a b ~ a2 b0 + a1 b1 + a0 b2 + a1 b0 + a0 b1 + a0 b0        (added in this order, small terms first),
// dropping terms of 2^-24 relative and below (the size of one fp32 rounding), on v_mfma_f32_32x32x16_bf16 with
// fp32 accumulators.  The accumulator chain is K / 16 long where the fp32 MFMA's is K / 2: measured against
// fp64 the result is slightly MORE accurate than the k-ordered fp32 fma chain (relative L2 error 4.8e-7 vs
// 5.7e-7 at K = 1024, profiles/r05_gemm_x6s_lab_a.txt).  Six bf16 MFMAs cost 6 / 16 of one fp32 MFMA product.
// Limits: |x| must stay below the bf16 overflow threshold (3.39e38) — an Inf / NaN operand gives NaN where
// the fp32 chain would give Inf / NaN.
//
// Data path.  Operands stay fp32 in memory (no plane tensors in HBM, no split launch — round 4's lab read
// pre-split planes: 6 bytes per element through L2 and stopped at 55 us on 4096 x 1024 x 1024).  A workgroup
// of 8 waves owns a 128 x 128 tile, k tile 32; every thread stages 8 floats of A and 8 of B per k tile, splits
// them (v_cvt_pk_bf16_f32 + exact residuals, 44 VALU per 8 floats) and writes the planes to LDS
// ([stage 2][operand 2][plane 3][row 128][80 B]: 32 bf16 + 16 B pad — the ds_read_b128 fragment reads are
// conflict-free).  A wave computes 64 x 32 = two 32x32 accumulators: 24 MFMAs per k tile in 24 "slots"; behind
// every MFMA a fixed share of the tile's other work is issued (pinned: see the anchors below):
//   first half  (k16 step 0): global loads of tile t+2 | B planes of t+1 (split in the previous half) -> LDS |
//                             split A of t+1 -> LDS | fragment reads of step 1
//   second half (k16 step 1): ONE barrier two slots in | split B of t+2 (registers) | fragment reads of
//                             step 0 of tile t+1
// Operand layouts: k-contiguous (x W^T forward) or row-contiguous (dX's W, both operands of dW).  A
// row-contiguous operand is loaded as 8 coalesced dwords per thread (one row, 8 consecutive k), so both kinds
// produce the same LDS image and no transposing read is needed.
// Edges: rows past M / N are read from clamped in-range addresses and only feed C rows / columns that are not
// stored; the K tail is zero-filled at load time.
#include "fx_common.h"
#include "fx_gemm_int.h"

#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 x6_bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 x6_bf16x2;
typedef __attribute__((ext_vector_type(2))) float x6_f32x2;
typedef __attribute__((ext_vector_type(4))) uint32_t x6_u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t x6_u32x2;

#define X6_ROWB 80
#define X6_PLANE (128 * X6_ROWB)
#define X6_OPER (3 * X6_PLANE)
#define X6_STAGE (2 * X6_OPER)
#define X6_LDS (2 * X6_STAGE)          // 122 880 B: one workgroup per CU

__device__ __forceinline__ uint32_t x6_pk_bf16(float a, float b) {     // v_cvt_pk_bf16_f32 (RNE), a in the low half
    x6_f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, x6_bf16x2));
}

// One operand's staging for a k tile: 8 floats per thread -> three planes of 4 packed dwords.
template <bool KC>
struct X6Opnd {
    static constexpr int NL = KC ? 2 : 8;     // global load instructions per tile
    static constexpr int NW = KC ? 6 : 3;     // LDS write instructions per tile
    const char* P;
    int64_t tstride, kstride;                 // bytes per k tile / per k (row-contiguous)
    int32_t ld, kbeg, kend, nk_full;
    int32_t rc[2];                            // KC: clamped row * ld of float4 p; else: clamped row
    int32_t kl;                               // k of this thread's first element inside a tile
    uint32_t voff[2];                         // byte offset of the load(s) in tile 0
    uint32_t woff[2];                         // LDS byte offset inside plane 0 of the operand image

    __device__ __forceinline__ void init(const float* P_, int64_t ld_, int64_t r0, int64_t Rext, int64_t kbeg_,
                                         int64_t kend_) {
        const int tid = threadIdx.x;
        P = reinterpret_cast<const char*>(P_);
        ld = (int32_t)ld_;
        kbeg = (int32_t)kbeg_;
        kend = (int32_t)kend_;
        nk_full = (kend - kbeg) / FX_BK;
        if constexpr (KC) {
            tstride = FX_BK * 4;
            kstride = 0;
            kl = (tid & 7) * 4;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int q = tid + 512 * p;
                // rows of a wave permuted (bits 0 and 2 swapped): the 16 lanes one ds_write_b64 lane group
                // serves hold rows R and R + 4 (80 dwords apart = 16 mod 32 banks) instead of R and R + 1 (20
                // apart: banks 0-3 hit twice — every plane write ran 2-way conflicted in the first cut)
                const int rl = q >> 3;
                const int row = (rl & ~5) | ((rl & 1) << 2) | ((rl >> 2) & 1);
                const int32_t r = (int32_t)r0 + row;
                rc[p] = (r < (int32_t)Rext ? r : (int32_t)Rext - 1) * ld;
                voff[p] = (uint32_t)(rc[p] + kbeg + kl) * 4u;
                woff[p] = (uint32_t)(row * X6_ROWB + (q & 7) * 8);
            }
        } else {
            tstride = (int64_t)FX_BK * ld * 4;
            kstride = (int64_t)ld * 4;
            const int row = tid & 127, o = tid >> 7;
            kl = 8 * o;
            const int32_t r = (int32_t)r0 + row;
            rc[0] = r < (int32_t)Rext ? r : (int32_t)Rext - 1;
            rc[1] = 0;
            voff[0] = (uint32_t)((kbeg + kl) * ld + rc[0]) * 4u;
            voff[1] = 0;
            woff[0] = (uint32_t)(row * X6_ROWB + o * 16);
            woff[1] = 0;
        }
    }

    // load instruction I of tile t.  MASK = false: the tile lies fully inside [kbeg, kend) — uniform tile base +
    // 32-bit per-lane byte offset (the saddr form); MASK = true (the slab's last tiles, the pipeline fill):
    // elements past kend are read from clamped in-range addresses here and zeroed WHERE THEY ARE CONSUMED
    // (mask_raw, called where the tile is split) — a select on the loaded registers right behind the load would wait for it and
    // drain the prefetch in the last iterations of every slab.
    template <int I, bool MASK>
    __device__ __forceinline__ void load_piece(int32_t t, float (&r)[8]) const {
        if constexpr (!MASK) {
            const char* b = P + (int64_t)t * tstride;
            // (the per-lane offset is "touched" inside the loop: hoisted out of it, its zero-extension to 64
            // bits lands in another basic block and the instruction selector no longer sees the
            // saddr + 32-bit voffset form — it then builds a 64-bit address per load in VGPRs that alias
            // in-flight fragment registers: lgkmcnt waits in front of address arithmetic)
            uint32_t vo = voff[KC ? I : 0];
            asm volatile("" : "+v"(vo));
            if constexpr (KC) {
                const float4 v = *reinterpret_cast<const float4*>(b + vo);
                r[4 * I + 0] = v.x; r[4 * I + 1] = v.y; r[4 * I + 2] = v.z; r[4 * I + 3] = v.w;
            } else {
                r[I] = *reinterpret_cast<const float*>(b + I * kstride + vo);
            }
        } else {
            if constexpr (KC) {
                const int32_t k = kbeg + t * FX_BK + kl;
                const int32_t kc = k < kend ? k : kend - 4;    // (K % 4 == 0: a float4 is in or out as a whole)
                const float4 v = *reinterpret_cast<const float4*>(P + (int64_t)(rc[I] + kc) * 4);
                r[4 * I + 0] = v.x; r[4 * I + 1] = v.y; r[4 * I + 2] = v.z; r[4 * I + 3] = v.w;
            } else {
                const int32_t k = kbeg + t * FX_BK + kl + I;
                const int32_t kc = k < kend ? k : kend - 1;
                r[I] = *reinterpret_cast<const float*>(P + ((int64_t)kc * ld + rc[0]) * 4);
            }
        }
    }

    // split sub-op S (0..11): pair j = S / 3, plane step st = S % 3; r is overwritten by the residuals.
    // The conversion is issued as a volatile asm statement: these are pure VALU operations that the instruction
    // selector is otherwise free to place anywhere between the global load and the LDS write (the first cut
    // floated all of them to the top of the half; a later one copied the freshly loaded registers right behind
    // their load — s_waitcnt vmcnt(0) there drains the prefetch, scripts/check_x6_isa.py).  asm volatile
    // statements keep their order relative to each other and to sched_barrier, the shifts / subtractions hang
    // off the conversion's result, so the work stays in the slot it was written in and the raw registers are
    // read where they are.  (s_nop 0: gfx950 wants one wait state between a VALU write and v_cvt_pk_bf16_f32
    // reading it — the compiler inserts it for its own conversions.)
    // The staged tile was tile t of the slab and may cross kend: its elements past kend become zero before
    // anything consumes them (the first conversion, the fused row sums) — see load_piece.
    __device__ __forceinline__ void mask_raw(float (&r)[8], int32_t t) const {
        const int32_t k0 = kbeg + t * FX_BK + kl;
        if constexpr (KC) {
            if (!(k0 < kend)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) r[e] = 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (!(k0 + e < kend)) r[e] = 0.f;
        }
    }

    template <int S>
    static __device__ __forceinline__ void split_piece(float (&r)[8], uint32_t (&pl)[3][4]) {
        constexpr int j = S / 3, st = S % 3;
        uint32_t p;
        asm volatile("s_nop 0\n\tv_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(p) : "v"(r[2 * j]), "v"(r[2 * j + 1]));
        pl[st][j] = p;
        if constexpr (st < 2) {
            r[2 * j] = r[2 * j] - __uint_as_float(p << 16);                    // exact
            r[2 * j + 1] = r[2 * j + 1] - __uint_as_float(p & 0xffff0000u);    // exact
        }
    }

    // LDS write W (0..NW-1) of the planes into the operand image at `dst` (plane 0)
    template <int W>
    __device__ __forceinline__ void write_piece(unsigned char* dst, const uint32_t (&pl)[3][4]) const {
        if constexpr (KC) {
            constexpr int g = W / 3, p = W % 3;     // float4 g (rows + 64 g), plane p
            x6_u32x2 v = {pl[p][2 * g], pl[p][2 * g + 1]};
            *reinterpret_cast<x6_u32x2*>(dst + p * X6_PLANE + woff[g]) = v;
        } else {
            x6_u32x4 v = {pl[W][0], pl[W][1], pl[W][2], pl[W][3]};
            *reinterpret_cast<x6_u32x4*>(dst + W * X6_PLANE + woff[0]) = v;
        }
    }
};

// One 128 x 128 output tile (linear tile index L of tiles_m x tiles_n, K slab z).  512 threads.
template <bool A_KC, bool B_KC>
__device__ __forceinline__ void fx_gemm_x6_tile(const GemmArgs& a, const int64_t L, const int z,
                                                unsigned char* const lds) {
    using OA = X6Opnd<A_KC>;
    using OB = X6Opnd<B_KC>;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int wm = w >> 2, wn = w & 3;
    const int l31 = lane & 31, half = lane >> 5;
    // XCD-aware tile order: workgroup b runs on XCD b % 8; every XCD gets a contiguous range of tiles
    // (row-major over (tm, tn)), so the n-tiles that share an A panel share one L2
    const int64_t nwg = (int64_t)a.tiles_m * a.tiles_n;
    int64_t T = L;
    if (nwg >= 8) {
        const int64_t q = nwg >> 3, r = nwg & 7, xcd = L & 7;
        T = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (L >> 3);
    }
    const int64_t m0 = (T / a.tiles_n) * 128, n0 = (T % a.tiles_n) * 128;
    const int64_t kbeg = (int64_t)z * a.k_chunk;
    const int64_t kend = (kbeg + a.k_chunk < a.K) ? kbeg + a.k_chunk : a.K;
    const int nk = (kend > kbeg) ? (int)((kend - kbeg + FX_BK - 1) / FX_BK) : 0;

    // fused row sums of op(A) (the bias gradient when op(A) = dZ^T, row-contiguous): every thread adds the 8
    // raw values of its row it stages per k tile; the four k-octet threads of a row meet in LDS at the end
    const bool do_rowsum = !A_KC && (a.epi.rowsum != nullptr) && (n0 == 0);
    float rs = 0.f;

    // Two accumulators per 32x32 block: the leading product a0 b0 on its own chain, the five correction
    // products (2^-8 and 2^-16 of it) on a second one.  Every MFMA rounds its accumulator once; with one chain
    // per block all six products of a k16 step rounded at the size of the full sum (6 K / 16 roundings: relative
    // L2 error 4.8e-7 at K = 1024, and the first-step gradients of c3's lower tower layers 3 x further from
    // fp64 than the fp32 chain's, tests/test_gpu_grad_parity.py) — now K / 16 roundings at full size and 5 K / 16
    // at 2^-8 of it; the chains meet in one addition at the end.  Four independent MFMA chains per wave.
    f32x16 acc[2], accs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[i][r] = 0.f; accs[i][r] = 0.f; }

    if (nk > 0) {
        OA oa;
        OB ob;
        oa.init(a.A, a.lda, m0, a.M, kbeg, kend);
        ob.init(a.B, a.ldb, n0, a.N, kbeg, kend);
        float ra[2][8], rb[8];
        uint32_t pa[3][4], pb[3][4];
        x6_bf16x8 fa[2][3][2], fb[2][3];       // [k16 step][plane][block]
        const uint32_t fo_a = (uint32_t)((wm * 64 + l31) * X6_ROWB + half * 16);
        const uint32_t fo_b = (uint32_t)(X6_OPER + (wn * 32 + l31) * X6_ROWB + half * 16);

        // fragment read R (0..8) of k16 step S from stage st, in the order of first use: b2 a0 | b1 a1 | b0 a2
        auto frag_read = [&](auto rr, auto ss, const unsigned char* st) {
            constexpr int R = decltype(rr)::value, S = decltype(ss)::value;
            if constexpr (R == 0) fb[S][2] = *reinterpret_cast<const x6_bf16x8*>(st + fo_b + 2 * X6_PLANE + S * 32);
            else if constexpr (R == 1) fa[S][0][0] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + S * 32);
            else if constexpr (R == 2) fa[S][0][1] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + 32 * X6_ROWB + S * 32);
            else if constexpr (R == 3) fb[S][1] = *reinterpret_cast<const x6_bf16x8*>(st + fo_b + X6_PLANE + S * 32);
            else if constexpr (R == 4) fa[S][1][0] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + X6_PLANE + S * 32);
            else if constexpr (R == 5) fa[S][1][1] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + X6_PLANE + 32 * X6_ROWB + S * 32);
            else if constexpr (R == 6) fb[S][0] = *reinterpret_cast<const x6_bf16x8*>(st + fo_b + S * 32);
            else if constexpr (R == 7) fa[S][2][0] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + 2 * X6_PLANE + S * 32);
            else fa[S][2][1] = *reinterpret_cast<const x6_bf16x8*>(st + fo_a + 2 * X6_PLANE + 32 * X6_ROWB + S * 32);
        };
        // MFMA m (0..11) of k16 step S: product m / 2, block m % 2.  Operands swapped (B first): the
        // accumulators hold the TRANSPOSED 32x32 block, a lane owns one row m of C and four adjacent columns
        // per register group — the 16-byte epilogue of fx_gemm_pipe_tile's TR form.  The empty asm pins the
        // MFMA to its slot (the builtin is a pure operation; its operands were read half a tile earlier).
        auto mfma = [&](auto mm, auto ss) {
            constexpr int m = decltype(mm)::value, S = decltype(ss)::value;
            constexpr int p = m / 2, i = m % 2;
            constexpr int bp = p == 0 ? 2 : (p == 1 || p == 3) ? 1 : 0;
            constexpr int ap = p == 0 ? 0 : p == 1 ? 1 : p == 2 ? 2 : p == 3 ? 0 : p == 4 ? 1 : 0;
            asm volatile("" : "+v"(fa[S][ap][i]));
            if constexpr (p == 5)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][bp], fa[S][ap][i], acc[i], 0, 0, 0);
            else
                accs[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[S][bp], fa[S][ap][i], accs[i], 0, 0, 0);
        };

        // one k tile; kt = -1 (DO = false) is the pipeline's fill: everything but the MFMAs and step-1 reads
        auto body = [&](int kt, auto par, auto domf, auto msk) {
            constexpr int P = decltype(par)::value;          // kt & 1
            constexpr bool DO = decltype(domf)::value;
            constexpr bool MASK = decltype(msk)::value;      // tile kt + 2 (loaded here) may cross kend
            unsigned char* const st_cur = lds + P * X6_STAGE;            // tile kt
            unsigned char* const st_nxt = lds + (P ^ 1) * X6_STAGE;      // tile kt + 1
            const int32_t tl = kt + 2 < nk ? kt + 2 : nk - 1;       // B is loaded and split in this iteration
            const int32_t tsa = kt + 1 < nk ? kt + 1 : nk - 1;      // the tile whose A is split here
            constexpr int NLB = OB::NL, NLA = OA::NL, NWB = OB::NW, NWA = OA::NW;
            // first half's list: [B loads][A loads][B plane writes][A split 0..5][A writes of float4 0 (KC)]
            //                    [A split 6..11][remaining A writes]
            constexpr int NWA1 = A_KC ? 3 : 0;
            constexpr int X1 = NLB, X2 = X1 + NLA, X3 = X2 + NWB, X4 = X3 + 6, X5 = X4 + NWA1, X6 = X5 + 6,
                          NX = X6 + (NWA - NWA1);
            auto xop = [&](auto ii) {
                constexpr int I = decltype(ii)::value;
                if constexpr (I < X1) ob.template load_piece<I, MASK>(tl, rb);
                else if constexpr (I < X2) oa.template load_piece<I - X1, MASK>(tl, ra[P]);
                else if constexpr (I < X3) ob.template write_piece<I - X2>(st_nxt + X6_OPER, pb);
                else if constexpr (I < X4) {
                    if constexpr (I == X3 && MASK) oa.mask_raw(ra[P ^ 1], tsa);
                    if constexpr (I == X3 && !A_KC) {
                        if (do_rowsum && kt + 1 < nk) {      // workgroup-uniform; tile kt + 1 is a real tile
                            const float (&r)[8] = ra[P ^ 1];
                            rs += ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                        }
                    }
                    OA::template split_piece<I - X3>(ra[P ^ 1], pa);
                } else if constexpr (I < X5) oa.template write_piece<I - X4>(st_nxt, pa);
                else if constexpr (I < X6) OA::template split_piece<6 + I - X5>(ra[P ^ 1], pa);
                else oa.template write_piece<I - X6 + NWA1>(st_nxt, pa);
            };
            fx_static_for<0, 12>([&](auto mm) {
                constexpr int m = decltype(mm)::value;
                if constexpr (DO) mfma(mm, std::integral_constant<int, 0>{});
                fx_static_for<(m * NX) / 12, ((m + 1) * NX) / 12>(xop);
                if constexpr (DO && m < 9) frag_read(mm, std::integral_constant<int, 1>{}, st_cur);
                __builtin_amdgcn_sched_barrier(0);
            });
            // second half.  The barrier sits two slots in: the plane writes and step-1 reads of the first half
            // have had those slots to land, so the lgkmcnt(0) in front of it is (nearly) free.  After it every
            // wave's planes of tile kt + 1 are in st_nxt and nobody reads st_cur any more (its last reads were
            // the step-1 fragments): the next tile's first half may overwrite it.
            fx_static_for<0, 12>([&](auto mm) {
                constexpr int m = decltype(mm)::value;
                if constexpr (m == 2) {
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (DO) mfma(mm, std::integral_constant<int, 1>{});
                if constexpr (m == 0 && MASK) ob.mask_raw(rb, tl);
                OB::template split_piece<m>(rb, pb);
                if constexpr (m >= 2 && m < 11)
                    frag_read(std::integral_constant<int, m - 2>{}, std::integral_constant<int, 0>{}, st_nxt);
                __builtin_amdgcn_sched_barrier(0);
            });
        };

        using P0 = std::integral_constant<int, 0>;
        using P1 = std::integral_constant<int, 1>;
        // fill: tile 0 raw -> rb / ra[0]; B planes of tile 0; then the MFMA-less pass for kt = -1
        using TT = std::true_type;
        using FF = std::false_type;
        fx_static_for<0, OB::NL>([&](auto ii) { ob.template load_piece<decltype(ii)::value, true>(0, rb); });
        fx_static_for<0, OA::NL>([&](auto ii) { oa.template load_piece<decltype(ii)::value, true>(0, ra[0]); });
        ob.mask_raw(rb, 0);
        fx_static_for<0, 12>([&](auto ss) { OB::template split_piece<decltype(ss)::value>(rb, pb); });
        body(-1, P1{}, FF{}, TT{});
        // tile kt + 2 is loaded in iteration kt: plain bodies while it has all 32 k inside the slab.  Pairs in
        // the loops, the odd tail outside (a skip path inside a loop would join two "loads in flight" states
        // at its back edge: fx_gemm_pipe_tile).
        const int n_plain = oa.nk_full - 2;
        int kt = 0;
        for (; kt + 1 < n_plain; kt += 2) {
            body(kt, P0{}, TT{}, FF{});
            body(kt + 1, P1{}, TT{}, FF{});
        }
        for (; kt + 1 < nk; kt += 2) {
            body(kt, P0{}, TT{}, TT{});
            body(kt + 1, P1{}, TT{}, TT{});
        }
        if (kt < nk) body(kt, P0{}, TT{}, TT{});
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] += accs[i][r];

    if (a.epi.rowsum != nullptr && n0 == 0 && !A_KC) {        // workgroup-uniform
        float* const red = reinterpret_cast<float*>(lds);
        __syncthreads();                                       // every fragment read of the last tile is done
        red[tid] = rs;                                         // [k octet][row]
        __syncthreads();
        if (tid < 128) {
            const float tot = (red[tid] + red[128 + tid]) + (red[256 + tid] + red[384 + tid]);
            const int64_t m = m0 + tid;
            if (m < a.M) {
                if (a.split_k > 1) a.ws[(int64_t)a.split_k * a.M * a.N + (int64_t)z * a.M + m] = tot;
                else a.epi.rowsum[m] = tot;
            }
        }
    }

    // epilogue (fx_gemm_pipe_tile's TR form): lane -> row m = l31 of a block; registers 4q .. 4q+3 ->
    // columns 8q + 4 half + 0..3
    const int64_t mb = m0 + wm * 64 + l31, nb = n0 + wn * 32 + 4 * half;
    if (a.split_k > 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = mb + i * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = nb + 8 * q;
                if (m < a.M && n < a.N)
                    *reinterpret_cast<float4*>(a.ws + ((int64_t)z * a.M + m) * a.N + n) =
                        make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]);
            }
        }
    } else {
        FxEpiOps4 ops[2][4];
        // (the operand loads of both blocks are issued before the first store: a load may not move above a
        // store to memory the compiler cannot prove distinct)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = mb + i * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = nb + 8 * q;
                if (m < a.M && n < a.N) fx_epi_load4(a.epi, m, n, ops[i][q]);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int64_t m = mb + i * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t n = nb + 8 * q;
                if (m < a.M && n < a.N) {
                    const float4 v = make_float4(acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2],
                                                 acc[i][4 * q + 3]);
                    *reinterpret_cast<float4*>(a.C + m * a.ldc + n) = fx_epi_apply4(a.epi, v, m, n, ops[i][q]);
                }
            }
        }
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_x6(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[X6_LDS];
    fx_gemm_x6_tile<A_KC, B_KC>(a, blockIdx.x, blockIdx.y, smem);
}

// Up to FX_MULTI_MAX independent GEMMs in one grid (fx_gemm_f32_batch): the dW and dX products of a layer, the
// cross and the deep layer of one DCNv2 depth.  Workgroups of the problems in `start` order.
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_gemm_x6_multi(MultiArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[X6_LDS];
    int i = 0;
    while (i + 1 < a.n && (int32_t)blockIdx.x >= a.start[i + 1]) ++i;
    // (arguments through the kernarg segment pointer: indexing the by-value struct with a run-time index
    // makes the compiler copy it to scratch — fx_gemm.hip, k_gemm_f32_multi)
    const MultiArgs* ka = (const MultiArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const GemmArgs& g = ka->p[i];
    int64_t L = (int64_t)blockIdx.x - a.start[i];
    const int64_t nt = (int64_t)g.tiles_m * g.tiles_n;
    const int z = (int)(L / nt);
    L -= (int64_t)z * nt;
    switch (ka->cfg[i] & 3) {
        case 0: fx_gemm_x6_tile<false, false>(g, L, z, smem); break;
        case 1: fx_gemm_x6_tile<true, false>(g, L, z, smem); break;
        case 2: fx_gemm_x6_tile<false, true>(g, L, z, smem); break;
        default: fx_gemm_x6_tile<true, true>(g, L, z, smem); break;
    }
}

bool fx_gemm_x6_enabled() {
    static const bool on = []() {
        const char* e = getenv("FX_GEMM_BF16X6");
        return !(e && atoi(e) == 0);
    }();
    return on;
}

int fx_gemm_x6_launch(bool a_kc, bool b_kc, const GemmArgs& a, hipStream_t s) {
    const dim3 grid((unsigned)((int64_t)a.tiles_m * a.tiles_n), (unsigned)a.split_k), block(512);
    if (a_kc && b_kc) hipLaunchKernelGGL((k_gemm_x6<true, true>), grid, block, 0, s, a);
    else if (a_kc) hipLaunchKernelGGL((k_gemm_x6<true, false>), grid, block, 0, s, a);
    else if (b_kc) hipLaunchKernelGGL((k_gemm_x6<false, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((k_gemm_x6<false, false>), grid, block, 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

int fx_gemm_x6_launch_multi(const MultiArgs& ma, int64_t workgroups, hipStream_t s) {
    hipLaunchKernelGGL(k_gemm_x6_multi, dim3((unsigned)workgroups), dim3(512), 0, s, ma);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
