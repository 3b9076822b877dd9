// fx_din_attn.hip — DIN target attention with the attention MLP fused in (second
// native version): the [B*L, 4E] concatenation, the [B*L, H] hidden activations and their gradients
// never exist in HBM.
//
// Reference (paths relative to the reference checkout):
//   fuxictr/pytorch/layers/attentions/target_attention.py:66-92   DIN_Attention.forward
//       x_{b,l} = [q_b, k_{b,l}, q_b - k_{b,l}, q_b * k_{b,l}]            (4E values per position)
//       a_{b,l} = W2 . Dice(W1 x_{b,l} + b1) + b2                         (MLP 4E -> H -> 1)
//   fuxictr/pytorch/layers/activations.py:24-51                    Dice
//       p = sigmoid(BatchNorm1d(z; affine=False)), y = p z + alpha (1 - p) z; the batch statistics
//       run over ALL B*L rows (padded positions included: the mask is applied after the MLP)
//   autograd of the above (rank_model.py:320)
//
// The first version (fx_din.hip + fx_gemm.hip) wrote the concatenation (52 MB at B = 4096, L = 50,
// E = 16), ran a 204800 x 64 x 64 GEMM that is nothing but prologue and epilogue, wrote the hidden
// tensor (52 MB), streamed it twice more for Dice and once for the 64 -> 1 head; backward the same
// again (profiles/r02_step_timeline_din.txt: 426 us of an 845 us step).  Dice needs the statistics
// of the whole batch before any row can be finished, so the fused form is two passes that RECOMPUTE
// the hidden layer from the 2 x 16 floats a position really has (q_b and k_{b,l}, 13 MB in total):
//   forward : stats pass  (h -> per-workgroup partial sums of h and h^2)          -> [host: finish /
//             all-reduce across ranks] -> apply pass (h -> Dice -> . W2 -> a[B*L] -> mask -> the
//             weighted sum over the sequence, out[B,E]: target_attention.py:85-91 rides along)
//   backward: sums pass   (da = mask (dout . k); h, da -> dalpha, sum dzhat, sum dzhat*zhat, dW2, db2)
//             -> [all-reduce] -> apply pass (h -> dh -> dW1 / db1 partials, dx -> dq, dK incl. the
//             pooling's share a mask dout)
// Every pass is one launch over 32-position tiles, one wave per tile stream:
//   * the tile's x^T [4E][32] is built in LDS from q / K rows read with 16-byte loads,
//   * the hidden layer of the tile on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32
//     products), in the orientation the pass needs — the operand order of the MFMA decides which
//     index lands in the lanes: h^T = W1 x^T puts ONE POSITION PER LANE and the hidden units in
//     registers (apply passes: Dice and the H -> 1 head are register loops, no cross-lane traffic);
//     h = x W1^T puts ONE UNIT PER LANE (statistics passes: the per-unit sums are register loops and
//     the unit's parameters lane constants),
//   * backward: dh goes through LDS once (row stride odd: conflict-free in both directions) and
//     feeds two more MFMA products, dW1 += dh^T x (accumulated in registers over all tiles of the
//     wave; db1 rides along on its A fragments) and dx^T = W1^T dh^T; dq is summed in position order
//     by the wave that owns the sample.
// All sums have a fixed order (register loops, fixed trees across half-waves / waves / workgroups):
// deterministic, hipGraph-replay bit-identical.
// Limits of the fused form: E <= 16 (4E <= 64), H <= 64, one hidden layer with Dice; everything else
// takes the unfused kernels (fx_din.hip) — both are native paths.
#include "fx_common.h"

#include <stdlib.h>

typedef float da_f32x16 __attribute__((ext_vector_type(16)));

#define DA_LDX 33   // row stride of the per-wave x^T tile [feature][position]; odd -> conflict-free
                    // when lanes walk positions (stride 1) AND when lanes walk features (stride 33)

struct DinAttnArgs {
    const float* q;
    int64_t q_ld;
    const float* K;
    int64_t k_ldb, k_ldl;
    int64_t n_rows;          // B * L positions
    int32_t nb;              // B
    int64_t rows_per_wave;   // contiguous positions a wave owns (fwd / bwd apply: whole samples)
    int32_t L, E, H;
    int32_t vec;             // q / K / dout / dK rows may be moved with 16-byte accesses
    const float* W1;         // [H, 4E]
    const float* b1;         // [H] or null
    const float* alpha;      // [H]
    const float* stats;      // mean[H], biased var[H]
    float eps;
    const float* W2;         // [H]
    const float* b2;         // [1] or null
    const int32_t* mask;     // [B, L] (row stride m_ld), position kept when != 0; null = all kept
    int64_t m_ld;
    const float* dout;       // [B, E] gradient of the pooled output
    int64_t dout_ld;
    const float* a_in;       // [B*L] attention logits of the forward (bwd apply)
    const float* da_in;      // [B*L] gradient of the logits (bwd apply; written by the sums pass)
    const float* sums;       // [H + n] = sum dzhat, [2H + n] = sum dzhat * zhat (training mode)
    float inv_n;
    float* a_out;            // [B*L]
    float* out;              // [B, E] pooled output
    int64_t out_ld;
    float* da_out;           // [B*L]
    float* dq;
    int64_t dq_ld;
    int32_t dq_acc;            // dq += instead of dq = (the row already holds another gradient share)
    float* dK;
    int64_t dk_ldb, dk_ldl;
    float* partial;          // per-workgroup partial sums
};

__device__ __forceinline__ int da_rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int NB, int FB, int WAVES, bool BWD, bool POOL>
struct DaSmem {
    static constexpr int HP = 32 * NB, FP = 32 * FB, LDW = HP + 1, LDH = HP + 1;
    float W1s[FP * LDW];                        // W1^T, [feature][hidden]: W1s[f * LDW + n] = W1[n][f]
    float4 PA[HP];                              // {b1, mean, rstd, alpha}
    float4 PB[HP];                              // {w2, mean(dzhat), mean(dzhat zhat), 0}
    float Xs[WAVES][FP * DA_LDX];               // x^T tile per wave
    static constexpr int HSZ = 32 * LDH > FP * DA_LDX ? 32 * LDH : FP * DA_LDX;
    float Hs[BWD ? WAVES : 1][BWD ? HSZ : 1];   // dh tile per wave, [position][hidden]; then dx^T
    float Ps[POOL ? WAVES : 1][POOL ? 16 * DA_LDX : 1];   // a mask k per wave, [feature][position]
};

// W1 and the per-unit parameters into LDS (padding = neutral values); ends with a barrier
template <class S>
__device__ __forceinline__ void da_load_params(S& sm, const DinAttnArgs& a, int nthreads) {
    const int KX = 4 * a.E, H = a.H;
    for (int i = threadIdx.x; i < S::FP * S::LDW; i += nthreads) sm.W1s[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < H * KX; i += nthreads) {
        const int n = i / KX, f = i - n * KX;
        sm.W1s[f * S::LDW + n] = a.W1[i];
    }
    for (int n = threadIdx.x; n < S::HP; n += nthreads) {
        float4 pa = make_float4(0.f, 0.f, 1.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < H) {
            pa.x = a.b1 ? a.b1[n] : 0.f;
            if (a.stats) {
                pa.y = a.stats[n];
                pa.z = rsqrtf(a.stats[H + n] + a.eps);
            }
            pa.w = a.alpha ? a.alpha[n] : 0.f;
            pb.x = a.W2 ? a.W2[n] : 0.f;
            if (a.sums) {
                pb.y = a.sums[H + n] * a.inv_n;
                pb.z = a.sums[2 * H + n] * a.inv_n;
            }
        }
        sm.PA[n] = pa;
        sm.PB[n] = pb;
    }
    __syncthreads();
}

// What one position brings, as the lanes hold it: lane (l31 = position, half) owns features
// [8 half, 8 half + 8) of its q, K (and dout) rows.  Loading (global) and writing the x^T tile (LDS)
// are separate steps so that the rows of tile t+1 are requested before the matrix products of tile t
// (the only HBM latency of a pass).  EC > 0: E is the compile-time constant EC and rows are 16-byte
// aligned (the BASELINE shape E = 16 and E = 8); EC = 0: any E <= 16, any alignment.
struct DaRows {
    float qv[8], kv[8], dov[8];
    uint32_t b;
    int l;
    float m;      // 1 = position kept, 0 = masked
};

__device__ __forceinline__ void da_ld8(const float* p, bool vec, int e0, int E, float (&v)[8]) {
    if (vec) {
        const float4 t0 = *reinterpret_cast<const float4*>(p);
        const float4 t1 = *reinterpret_cast<const float4*>(p + 4);
        v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
        v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
    } else {
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e0 + u < E) v[u] = p[u];
    }
}

template <int EC, bool DOUT, bool MASK>
__device__ __forceinline__ void da_load_rows(const DinAttnArgs& a, int64_t row, bool valid, int half,
                                             DaRows& x) {
    const int E = EC ? EC : a.E, e0 = 8 * half;
    const bool vec = EC ? true : (a.vec != 0);
    x.b = 0;
    x.l = 0;
    x.m = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        x.qv[u] = 0.f;
        x.kv[u] = 0.f;
        x.dov[u] = 0.f;
    }
    if (valid) {
        const uint32_t r32 = (uint32_t)row, L32 = (uint32_t)a.L;     // B * L < 2^31 (checked on the host)
        x.b = r32 / L32;
        x.l = (int)(r32 - x.b * L32);
        if (MASK) x.m = (!a.mask || a.mask[(int64_t)x.b * a.m_ld + x.l] != 0) ? 1.f : 0.f;
        if (e0 < E) {
            da_ld8(a.q + (int64_t)x.b * a.q_ld + e0, vec, e0, E, x.qv);
            da_ld8(a.K + (int64_t)x.b * a.k_ldb + (int64_t)x.l * a.k_ldl + e0, vec, e0, E, x.kv);
            if (DOUT) da_ld8(a.dout + (int64_t)x.b * a.dout_ld + e0, vec, e0, E, x.dov);
        }
    }
}

// x^T[f][position] = [q, k, q - k, q * k] of the tile (rows outside the range were loaded as 0)
template <int EC>
__device__ __forceinline__ void da_store_x(float* Xs, int Erun, int l31, int half, const DaRows& x) {
    const int E = EC ? EC : Erun, e0 = 8 * half;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = e0 + u;
        if (e < E) {
            Xs[e * DA_LDX + l31] = x.qv[u];
            Xs[(E + e) * DA_LDX + l31] = x.kv[u];
            Xs[(2 * E + e) * DA_LDX + l31] = x.qv[u] - x.kv[u];
            Xs[(3 * E + e) * DA_LDX + l31] = x.qv[u] * x.kv[u];
        }
    }
}

// sigmoid on the transcendental unit: v_exp_f32 + v_rcp_f32 (1 ulp each) instead of the ~40
// instructions of expf + IEEE division — the Dice gate is evaluated 32 x per lane and tile, which
// made the passes VALU-bound (profiles/r02_step_timeline_din_fused_v1.txt); |error| <= 2e-7
__device__ __forceinline__ float da_sigmoid(float z) {
    return __builtin_amdgcn_rcpf(1.f + __expf(-z));
}

// The hidden layer of the wave's 32 positions on the matrix cores, bias not yet added.
//   POS_IN_LANE = true : h^T = W1 x^T -> acc[j][r] = h(position l31, unit 32 j + rowmap(r, half)):
//                        one position per lane, the units in registers (per-position reductions,
//                        e.g. the H -> 1 head, are register loops)
//   POS_IN_LANE = false: h = x W1^T   -> acc[j][r] = h(position rowmap(r, half), unit 32 j + l31):
//                        one unit per lane (per-unit statistics are register loops, the unit's
//                        parameters are lane constants)
// Same LDS reads either way; only the operand order of the MFMA changes.  KXC > 0 (= 4 EC): fully
// unrolled, the fragments of step k+1 are read before the MFMAs of step k are issued — the first
// version (a rolled loop: read, wait, 2 MFMAs, read, wait ...) exposed the LDS latency 32 x per tile.
template <int NB, bool POS_IN_LANE, int KXC, int LDW>
__device__ __forceinline__ void da_gemm_h(const float* __restrict__ W1s, const float* __restrict__ Xs,
                                          int KX, int l31, int half, da_f32x16 (&acc)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* xb = Xs + half * DA_LDX + l31;
    const float* wa = W1s + half * LDW + l31;
    if constexpr (KXC > 0) {
        constexpr int NK = KXC / 2;
        float bx[2], aw[2][NB];
        bx[0] = xb[0];
#pragma unroll
        for (int j = 0; j < NB; ++j) aw[0][j] = wa[32 * j];
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int c = kk & 1, n = c ^ 1;
            if (kk + 1 < NK) {
                bx[n] = xb[2 * (kk + 1) * DA_LDX];
#pragma unroll
                for (int j = 0; j < NB; ++j) aw[n][j] = wa[2 * (kk + 1) * LDW + 32 * j];
            }
            // pin the order "reads of step k+1, then MFMAs of step k": left alone, the scheduler
            // sinks every read to just before its MFMA (read, wait, 2 MFMAs, read, wait, ...) and
            // the matrix pipe idles for one LDS latency per step
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if constexpr (POS_IN_LANE)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[c][j], bx[c], acc[j], 0, 0, 0);
                else
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bx[c], aw[c][j], acc[j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const int nk = KX >> 1;
#pragma unroll 4
        for (int kk = 0; kk < nk; ++kk) {
            const float bx = xb[2 * kk * DA_LDX];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float aw = wa[2 * kk * LDW + 32 * j];
                if constexpr (POS_IN_LANE)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bx, acc[j], 0, 0, 0);
                else
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bx, aw, acc[j], 0, 0, 0);
            }
        }
    }
}

// Per-sample sums over the positions of a tile T[feature][position] (E rows x DA_LDX), carried across
// the tiles of a wave that owns whole samples: run (+)= the tile's positions of the current sample;
// when the sample's L-th position lies in the tile, out[b][e] = run and the rest of the tile opens the
// next sample.  L >= 32 (at most one sample ends inside a tile): lane (e = lane & 15, g = lane >> 4)
// adds positions [8 g, 8 g + 8), the four groups are combined by two xor-shuffles (every lane of a
// feature ends with the same bits); smaller L: lane e walks the positions in order.
__device__ __forceinline__ void da_seg_sum(const float* T, int E, int L, int lane, int nvalid,
                                           uint32_t& bcur, int& lcur, float& run, float* out,
                                           int64_t out_ld, float extra = 0.f, bool acc = false) {
    if (L >= 32) {
        const int e = lane & 15, g = lane >> 4;
        const int c = (L - lcur < nvalid) ? L - lcur : nvalid;   // positions that belong to sample bcur
        float v0 = 0.f, v1 = 0.f;
        if (e < E) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = T[e * DA_LDX + 8 * g + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = 8 * g + u;
                if (i < c) v0 += t[u];
                else if (i < nvalid) v1 += t[u];
            }
        }
        v0 += __shfl_xor(v0, 16, 64);
        v0 += __shfl_xor(v0, 32, 64);
        v1 += __shfl_xor(v1, 16, 64);
        v1 += __shfl_xor(v1, 32, 64);
        run += v0;
        if (lcur + c == L) {
            if (lane < E) {              // extra: per-sample term; acc: added to what the row holds
                float* o = out + (int64_t)bcur * out_ld + lane;
                *o = acc ? *o + (run + extra) : run + extra;
            }
            run = v1;
            ++bcur;
            lcur = nvalid - c;
        } else {
            lcur += c;
        }
    } else {
        for (int i = 0; i < nvalid; ++i) {
            if (lane < E) run += T[lane * DA_LDX + i];
            if (++lcur == L) {
                if (lane < E) {
                    float* o = out + (int64_t)bcur * out_ld + lane;
                    *o = acc ? *o + run : run;
                }
                run = 0.f;
                lcur = 0;
                ++bcur;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward, pass 1: per-workgroup partial sums of h and h^2 over the positions
//   partial[(wg * 2 + k) * H + n]
// ---------------------------------------------------------------------------------------------
template <int NB, int FB, int EC>
__global__ __launch_bounds__(256) void k_din_attn_stats(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false, false>;
    __shared__ S sm;
    da_load_params(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    const int KX = 4 * a.E, H = a.H;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    float s1[NB], s2[NB], b1v[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        s1[j] = 0.f;
        s2[j] = 0.f;
        b1v[j] = sm.PA[32 * j + l31].x;
    }
    DaRows cur;
    da_load_rows<EC, false, false>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        da_store_x<EC>(Xs, a.E, l31, half, cur);
        da_load_rows<EC, false, false>(a, rb + 32 + l31, rb + 32 + l31 < R1, half, cur);   // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, false, 4 * EC, S::LDW>(sm.W1s, Xs, KX, l31, half, acc);
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float z = acc[j][r] + b1v[j];
                if (da_rowmap(r, half) < nvalid) {
                    s1[j] += z;
                    s2[j] = fmaf(z, z, s2[j]);
                }
            }
    }
    __syncthreads();                               // every wave is done with its x tile
    float* red = &sm.Xs[0][0];                     // [4 waves][2][HP]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float o1 = __shfl_xor(s1[j], 32, 64), o2 = __shfl_xor(s2[j], 32, 64);
        if (half == 0) {                           // positions of half 0 first
            red[(wave * 2 + 0) * S::HP + 32 * j + l31] = s1[j] + o1;
            red[(wave * 2 + 1) * S::HP + 32 * j + l31] = s2[j] + o2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * S::HP) {
        const int k = threadIdx.x / S::HP, n = threadIdx.x % S::HP;
        if (n < H)
            a.partial[((int64_t)blockIdx.x * 2 + k) * H + n] =
                (red[(0 * 2 + k) * S::HP + n] + red[(1 * 2 + k) * S::HP + n]) +
                (red[(2 * 2 + k) * S::HP + n] + red[(3 * 2 + k) * S::HP + n]);
    }
}

// ---------------------------------------------------------------------------------------------
// forward, pass 2: a[b*L + l] = W2 . Dice(W1 x + b1) + b2;  out[b] = sum_l a mask k   (a wave owns
// whole samples)
// ---------------------------------------------------------------------------------------------
template <int NB, int FB, int EC>
__global__ __launch_bounds__(256) void k_din_attn_fwd(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false, true>;
    __shared__ S sm;
    da_load_params(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    float* Ps = sm.Ps[wave];
    const int E = EC ? EC : a.E, KX = 4 * E, L = a.L, e0 = 8 * half;
    const float b2 = a.b2 ? a.b2[0] : 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float run = 0.f;
    DaRows cur;
    da_load_rows<EC, false, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const float m = cur.m;
        da_store_x<EC>(Xs, E, l31, half, cur);
        da_load_rows<EC, false, true>(a, row + 32, row + 32 < R1, half, cur);              // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, true, 4 * EC, S::LDW>(sm.W1s, Xs, KX, l31, half, acc);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * j + da_rowmap(r, half);
                const float4 pa = sm.PA[n];
                const float w2 = sm.PB[n].x;
                const float z = acc[j][r] + pa.x;
                const float zh = (z - pa.y) * pa.z;
                const float p = da_sigmoid(zh);
                const float y = p * z + pa.w * (1.f - p) * z;
                t += y * w2;
            }
        const float o = __shfl_xor(t, 32, 64);
        const float ai = (half == 0 ? t + o : o + t) + b2;   // units of half 0 first, in both lanes
        if (half == 0 && valid) a.a_out[row] = ai;
        // pooled output: this position's share a mask k, summed per sample
        const float wm = valid ? ai * m : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u;
            if (e < E) Ps[e * DA_LDX + l31] = wm * Xs[(E + e) * DA_LDX + l31];
        }
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        da_seg_sum(Ps, E, L, lane, nvalid, bcur, lcur, run, a.out, a.out_ld);
    }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 1: da = mask (dout . k) per position (written for pass 2), and the per-workgroup
// partial sums over the positions
//   k = 0: dalpha[n]  1: sum dzhat[n]  2: sum dzhat zhat[n]  3: dW2[n]  4: db2 (at n = 0)
//   partial[(wg * 5 + k) * H + n]
// ---------------------------------------------------------------------------------------------
template <int NB, int FB, int EC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_din_attn_bwd_sums(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false, false>;
    __shared__ S sm;
    __shared__ float das[4][32];                   // da of the wave's 32 positions
    da_load_params(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    const int KX = 4 * a.E, H = a.H;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    float sa[NB], sd[NB], sz[NB], sw[NB];
    float4 pa[NB];
    float w2[NB];
    float sb2 = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        sa[j] = 0.f;
        sd[j] = 0.f;
        sz[j] = 0.f;
        sw[j] = 0.f;
        pa[j] = sm.PA[32 * j + l31];
        w2[j] = sm.PB[32 * j + l31].x;
    }
    DaRows cur;
    da_load_rows<EC, true, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        da_store_x<EC>(Xs, a.E, l31, half, cur);
        // da = mask * sum_e dout[b][e] k[e]  (fx_din_pool_bwd's dw): 8 features per lane, halves added
        float d = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) d = fmaf(cur.dov[u], cur.kv[u], d);
        const float od = __shfl_xor(d, 32, 64);
        const float da_i = (half == 0 ? d + od : od + d) * cur.m;      // 0 past the range (m = 0)
        if (half == 0) {
            das[wave][l31] = da_i;
            sb2 += da_i;
            if (valid) a.da_out[row] = da_i;
        }
        da_load_rows<EC, true, true>(a, row + 32, row + 32 < R1, half, cur);                // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, false, 4 * EC, S::LDW>(sm.W1s, Xs, KX, l31, half, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dar = das[wave][da_rowmap(r, half)];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float z = acc[j][r] + pa[j].x;
                const float zh = (z - pa[j].y) * pa[j].z;
                const float p = da_sigmoid(zh);
                const float y = p * z + pa[j].w * (1.f - p) * z;
                const float dy = dar * w2[j];
                const float dzh = dy * z * (1.f - pa[j].w) * p * (1.f - p);
                sa[j] = fmaf(dy * (1.f - p), z, sa[j]);
                sd[j] += dzh;
                sz[j] = fmaf(dzh, zh, sz[j]);
                sw[j] = fmaf(dar, y, sw[j]);
            }
        }
    }
    sb2 = fx_wave_sum(sb2);                         // lanes of half 1 hold 0
    __syncthreads();
    float* red = &sm.Xs[0][0];                     // [4 waves][4][HP] + [4] for db2
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float oa = __shfl_xor(sa[j], 32, 64), od = __shfl_xor(sd[j], 32, 64);
        const float oz = __shfl_xor(sz[j], 32, 64), ow = __shfl_xor(sw[j], 32, 64);
        if (half == 0) {
            red[(wave * 4 + 0) * S::HP + 32 * j + l31] = sa[j] + oa;
            red[(wave * 4 + 1) * S::HP + 32 * j + l31] = sd[j] + od;
            red[(wave * 4 + 2) * S::HP + 32 * j + l31] = sz[j] + oz;
            red[(wave * 4 + 3) * S::HP + 32 * j + l31] = sw[j] + ow;
        }
    }
    if (lane == 0) red[16 * S::HP + wave] = sb2;
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * S::HP; t += 256) {
        const int k = t / S::HP, n = t % S::HP;
        if (n < H)
            a.partial[((int64_t)blockIdx.x * 5 + k) * H + n] =
                (red[(0 * 4 + k) * S::HP + n] + red[(1 * 4 + k) * S::HP + n]) +
                (red[(2 * 4 + k) * S::HP + n] + red[(3 * 4 + k) * S::HP + n]);
    }
    for (int n = threadIdx.x; n < H; n += 256)
        a.partial[((int64_t)blockIdx.x * 5 + 4) * H + n] =
            n == 0 ? (red[16 * S::HP + 0] + red[16 * S::HP + 1]) +
                         (red[16 * S::HP + 2] + red[16 * S::HP + 3])
                   : 0.f;
}

// ---------------------------------------------------------------------------------------------
// backward, pass 2: dh -> dW1 / db1 partials (per workgroup), dq, dK (attention part + the pooling's
// share a mask dout)
//   partial[wg * (H * 4E + H) + n * 4E + f]  and  [... + H * 4E + n]
// A wave owns whole samples (rows_per_wave is a multiple of L), so dq[b] is finished inside the wave.
// ---------------------------------------------------------------------------------------------
template <int NB, int FB, int EC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_din_attn_bwd(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 2, true, false>;
    __shared__ S sm;
    da_load_params(sm, a, 128);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    float* Hs = sm.Hs[wave];
    const int E = EC ? EC : a.E, KX = 4 * E, H = a.H, L = a.L;
    const int e0 = 8 * half;
    const bool vec = EC ? true : (a.vec != 0);
    const int64_t gw = (int64_t)blockIdx.x * 2 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    // features [4E, FP) of the x tile are read by the dW1 product: zero once
    for (int i = lane; i < (S::FP - KX) * DA_LDX; i += 64) Xs[KX * DA_LDX + i] = 0.f;
    da_f32x16 accW[NB][FB];
    float db1q[NB];                                // unit 32 j + l31, positions of this half's parity
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int jb = 0; jb < FB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[j][jb][r] = 0.f;
        db1q[j] = 0.f;
    }
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float dq_run = 0.f;
    DaRows cur;
    da_load_rows<EC, false, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    float da_n = (R0 + l31 < R1) ? a.da_in[R0 + l31] : 0.f;
    float a_n = (R0 + l31 < R1) ? a.a_in[R0 + l31] : 0.f;
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const uint32_t b = cur.b;
        const int l = cur.l;
        const float da_i = da_n;
        const float wm = a_n * cur.m;                              // a mask (fx_din_pool_bwd's w m)
        da_store_x<EC>(Xs, E, l31, half, cur);
        da_load_rows<EC, false, true>(a, row + 32, row + 32 < R1, half, cur);              // tile t+1
        da_n = (row + 32 < R1) ? a.da_in[row + 32] : 0.f;
        a_n = (row + 32 < R1) ? a.a_in[row + 32] : 0.f;
        {
            da_f32x16 acc[NB];
            da_gemm_h<NB, true, 4 * EC, S::LDW>(sm.W1s, Xs, KX, l31, half, acc);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * j + da_rowmap(r, half);
                    const float4 pa = sm.PA[n];
                    const float4 pb = sm.PB[n];
                    const float z = acc[j][r] + pa.x;
                    const float zh = (z - pa.y) * pa.z;
                    const float p = da_sigmoid(zh);
                    const float dy = da_i * pb.x;
                    float dzh = dy * z * (1.f - pa.w) * p * (1.f - p);
                    dzh -= pb.y + zh * pb.z;                 // 0 outside training mode
                    float dh = dy * (p + pa.w * (1.f - p)) + dzh * pa.z;
                    if (!valid) dh = 0.f;
                    Hs[l31 * S::LDH + n] = dh;
                }
        }
        // dW1[n][f] += sum_i dh[i][n] x[i][f]  and  dx^T[f][i] = sum_n W1[n][f] dh[i][n], interleaved
        // (six independent accumulator chains keep the matrix pipe busy while fragments are read)
        da_f32x16 accD[FB];
#pragma unroll
        for (int jb = 0; jb < FB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accD[jb][r] = 0.f;
        {
            // fragments of step k+1 are read before the MFMAs of step k (order pinned as in da_gemm_h)
            float av[2][NB], bv[2][FB], bh[2][NB], aw[2][NB][FB];
            auto rd = [&](int kk, int s_) {
#pragma unroll
                for (int j = 0; j < NB; ++j) av[s_][j] = Hs[(2 * kk + half) * S::LDH + 32 * j + l31];
#pragma unroll
                for (int jb = 0; jb < FB; ++jb) bv[s_][jb] = Xs[(32 * jb + l31) * DA_LDX + 2 * kk + half];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int kd = kk * NB + t;     // HP / 2 = 16 NB steps over the hidden units
                    bh[s_][t] = Hs[l31 * S::LDH + 2 * kd + half];
#pragma unroll
                    for (int jb = 0; jb < FB; ++jb)
                        aw[s_][t][jb] = sm.W1s[(32 * jb + l31) * S::LDW + 2 * kd + half];
                }
            };
            rd(0, 0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int c = kk & 1;
                if (kk + 1 < 16) rd(kk + 1, c ^ 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    db1q[j] += av[c][j];           // db1 rides along: the A fragments ARE dh
#pragma unroll
                    for (int jb = 0; jb < FB; ++jb)
                        accW[j][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][j], bv[c][jb],
                                                                           accW[j][jb], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t)
#pragma unroll
                    for (int jb = 0; jb < FB; ++jb)
                        accD[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[c][t][jb], bh[c][t],
                                                                        accD[jb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // this sample's dout features of the lane (L1 / L2 hits: a tile spans 1-2 samples)
        float dov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dov[u] = 0.f;
        if (valid && e0 < E) da_ld8(a.dout + (int64_t)b * a.dout_ld + e0, vec, e0, E, dov);
        // dx^T through LDS, over the dh tile (both products have consumed it); the x tile still holds
        // this tile's q and k rows
#pragma unroll
        for (int jb = 0; jb < FB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Hs[(32 * jb + da_rowmap(r, half)) * DA_LDX + l31] = accD[jb][r];
        // dk = dx_k - dx_(q-k) + dx_(q*k) q + a mask dout ;  this position's share of
        // dq = dx_q + dx_(q-k) + dx_(q*k) k  (written over the (q-k) rows of the x tile, which nobody
        // reads any more)
        float dkv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u;
            dkv[u] = 0.f;
            if (e < E) {
                const float qe = Xs[e * DA_LDX + l31], ke = Xs[(E + e) * DA_LDX + l31];
                const float dxa = Hs[e * DA_LDX + l31];
                const float dxb = Hs[(E + e) * DA_LDX + l31];
                const float dxc = Hs[(2 * E + e) * DA_LDX + l31];
                const float dxd = Hs[(3 * E + e) * DA_LDX + l31];
                dkv[u] = (dxb - dxc + dxd * qe) + wm * dov[u];
                const float dqc = dxa + dxc + dxd * ke;
                Xs[(2 * E + e) * DA_LDX + l31] = valid ? dqc : 0.f;
            }
        }
        if (valid && e0 < E) {
            float* dkp = a.dK + (int64_t)b * a.dk_ldb + (int64_t)l * a.dk_ldl + e0;
            if (vec) {
                *reinterpret_cast<float4*>(dkp) = make_float4(dkv[0], dkv[1], dkv[2], dkv[3]);
                *reinterpret_cast<float4*>(dkp + 4) = make_float4(dkv[4], dkv[5], dkv[6], dkv[7]);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < E) dkp[u] = dkv[u];
            }
        }
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        da_seg_sum(Xs + 2 * E * DA_LDX, E, L, lane, nvalid, bcur, lcur, dq_run, a.dq, a.dq_ld, 0.f,
                   a.dq_acc != 0);
    }
    __syncthreads();                               // both waves have left their tile loops
    float* scratch = &sm.Xs[0][0];                 // wave 1's dW1 accumulators: NB*FB*16*64 floats
    float* bsc = &sm.Hs[0][0];                     // [2 waves][HP]
    if (wave == 1) {
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int jb = 0; jb < FB; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    scratch[((j * FB + jb) * 16 + r) * 64 + lane] = accW[j][jb][r];
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float o = __shfl_xor(db1q[j], 32, 64);
        if (half == 0) bsc[wave * S::HP + 32 * j + l31] = db1q[j] + o;   // even positions first
    }
    __syncthreads();
    if (wave == 0) {
        float* P = a.partial + (int64_t)blockIdx.x * ((int64_t)H * KX + H);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int jb = 0; jb < FB; ++jb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * j + da_rowmap(r, half), f = 32 * jb + l31;
                    if (n < H && f < KX)
                        P[(int64_t)n * KX + f] =
                            accW[j][jb][r] + scratch[((j * FB + jb) * 16 + r) * 64 + lane];
                }
        if (lane < S::HP && lane < H) P[(int64_t)H * KX + lane] = bsc[lane] + bsc[S::HP + lane];
    }
}

// =============================================================================================
// Second formulation ("q split") of the same four passes, used for the BASELINE shape (E = 8 / 16 with
// 16-byte rows, L >= 32).  With x = (q, k, q - k, q * k) and W1 = (Wa | Wb | Wc | Wd):
//     W1 x = (Wa + Wc) q  +  (Wb - Wc) k + Wd (q * k)
// The first term is the same for all L positions of a sample: hq[b] = (Wa + Wc) q_b + b1 costs 16 FMAs
// per hidden unit and SAMPLE, and what is left per POSITION is a 2E-wide product — half the MFMA
// work of every pass, half the x tile, half the LDS copy of W1 (so twice the waves fit a CU: the
// passes are serial chains per wave and live on having a second wave to switch to).  Backward:
//     dWa = sum_b dhs_b q_b^T,  dWb = sum dh k^T,  dWc = dWa - dWb,  dWd = sum dh (q*k)^T
//     dk  = (Wb - Wc)^T dh + (Wd^T dh) * q (+ the pooling's share),
//     dq_b = (Wa + Wc)^T dhs_b + sum_l (Wd^T dh_l) * k_l,      dhs_b = sum_l dh_{b,l}
// (dhs rides along on the A fragments of the dW product, like db1).  The sums are re-associated with
// respect to the reference's fp32 order (W1 x as one 4E-long dot product): differences ~1e-7
// relative, inside the parity bounds of tests/test_gpu_din_attn.py.  L >= 32 guarantees that a
// 32-position tile touches at most two samples.
// =============================================================================================
template <int NB, int WAVES, bool BWD, bool POOL>
struct Da2Smem {
    static constexpr int HP = 32 * NB, LDW = HP + 1, LDH = HP + 1;
    float Ws[32 * LDW];                  // rows e < E: (Wb - Wc)[n][e]; rows E + e: Wd[n][e]; then 0
    float Wqs[16 * LDW];                 // (Wa + Wc)^T: Wqs[e * LDW + n]
    float4 PA[HP];
    float4 PB[HP];
    float Xs[WAVES][32 * DA_LDX];        // x'^T tile per wave: rows e: k, rows E + e: q * k, then 0
    float Hq[WAVES][2 * HP];             // hq of the (at most) two samples of the current tile
    static constexpr int HSZ = 32 * LDH > HP * DA_LDX ? 32 * LDH : HP * DA_LDX;
    float Hs[BWD ? WAVES : 1][BWD ? HSZ : 1];
    float Ps[POOL ? WAVES : 1][POOL ? 16 * DA_LDX : 1];
};

template <class S, int EC>
__device__ __forceinline__ void da2_load_params(S& sm, const DinAttnArgs& a, int nthreads) {
    const int H = a.H;
    for (int i = threadIdx.x; i < 32 * S::LDW; i += nthreads) sm.Ws[i] = 0.f;
    for (int i = threadIdx.x; i < 16 * S::LDW; i += nthreads) sm.Wqs[i] = 0.f;
    __syncthreads();
    for (int i = threadIdx.x; i < H * EC; i += nthreads) {
        const int n = i / EC, e = i - n * EC;
        const float* w = a.W1 + (int64_t)n * 4 * EC + e;
        const float wa = w[0], wb = w[EC], wc = w[2 * EC], wd = w[3 * EC];
        sm.Ws[e * S::LDW + n] = wb - wc;
        sm.Ws[(EC + e) * S::LDW + n] = wd;
        sm.Wqs[e * S::LDW + n] = wa + wc;
    }
    for (int n = threadIdx.x; n < S::HP; n += nthreads) {
        float4 pa = make_float4(0.f, 0.f, 1.f, 0.f), pb = make_float4(0.f, 0.f, 0.f, 0.f);
        if (n < H) {
            pa.x = a.b1 ? a.b1[n] : 0.f;
            if (a.stats) {
                pa.y = a.stats[n];
                pa.z = rsqrtf(a.stats[H + n] + a.eps);
            }
            pa.w = a.alpha ? a.alpha[n] : 0.f;
            pb.x = a.W2 ? a.W2[n] : 0.f;
            if (a.sums) {
                pb.y = a.sums[H + n] * a.inv_n;
                pb.z = a.sums[2 * H + n] * a.inv_n;
            }
        }
        sm.PA[n] = pa;
        sm.PB[n] = pb;
    }
    __syncthreads();
}

template <int EC>
__device__ __forceinline__ void da2_store_x(float* Xs, int l31, int half, const DaRows& x) {
    const int e0 = 8 * half;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int e = e0 + u;
        if (e < EC) {
            Xs[e * DA_LDX + l31] = x.kv[u];
            Xs[(EC + e) * DA_LDX + l31] = x.qv[u] * x.kv[u];
        }
    }
}

// hq[s][j] = b1[n] + sum_e (Wa + Wc)[n][e] q[b0 + s][e] for the lane's units n = 32 j + l31, s = 0, 1
template <int NB, int EC, class S>
__device__ __forceinline__ void da2_hq(const S& sm, const DinAttnArgs& a, uint32_t b0, int l31,
                                       float (&hq)[2][NB]) {
#pragma unroll
    for (int s_ = 0; s_ < 2; ++s_) {
        const uint32_t b = b0 + s_;
        const bool ok = b < (uint32_t)a.nb;
        const float* qp = a.q + (int64_t)(ok ? b : 0) * a.q_ld;
#pragma unroll
        for (int j = 0; j < NB; ++j) hq[s_][j] = sm.PA[32 * j + l31].x;
#pragma unroll
        for (int e = 0; e < EC; ++e) {
            const float qe = ok ? qp[e] : 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                hq[s_][j] = fmaf(sm.Wqs[e * S::LDW + 32 * j + l31], qe, hq[s_][j]);
        }
    }
}

// sample bookkeeping of a wave that owns whole samples, L >= 32: c = positions of the tile that
// belong to sample bcur (the rest opens bcur + 1)
__device__ __forceinline__ int da2_cut(int nvalid, int L, int lcur) {
    return (L - lcur < nvalid) ? L - lcur : nvalid;
}
__device__ __forceinline__ void da2_advance(int nvalid, int L, int c, uint32_t& bcur, int& lcur) {
    if (lcur + c == L) {
        ++bcur;
        lcur = nvalid - c;
    } else {
        lcur += c;
    }
}

template <int NB, int EC>
__global__ __launch_bounds__(256) void k_din_attn2_stats(DinAttnArgs a) {
    using S = Da2Smem<NB, 4, false, false>;
    __shared__ S sm;
    da2_load_params<S, EC>(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    const int H = a.H, L = a.L;
    for (int i = lane; i < (32 - 2 * EC) * DA_LDX; i += 64) Xs[2 * EC * DA_LDX + i] = 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float s1[NB], s2[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        s1[j] = 0.f;
        s2[j] = 0.f;
    }
    DaRows cur;
    da_load_rows<EC, false, false>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        const int c = da2_cut(nvalid, L, lcur);
        float hq[2][NB];
        da2_hq<NB, EC>(sm, a, bcur, l31, hq);
        da2_store_x<EC>(Xs, l31, half, cur);
        da_load_rows<EC, false, false>(a, rb + 32 + l31, rb + 32 + l31 < R1, half, cur);   // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, false, 2 * EC, S::LDW>(sm.Ws, Xs, 2 * EC, l31, half, acc);
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = da_rowmap(r, half);
                const float z = acc[j][r] + (i < c ? hq[0][j] : hq[1][j]);
                if (i < nvalid) {
                    s1[j] += z;
                    s2[j] = fmaf(z, z, s2[j]);
                }
            }
        da2_advance(nvalid, L, c, bcur, lcur);
    }
    __syncthreads();
    float* red = &sm.Xs[0][0];                     // [4 waves][2][HP]
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float o1 = __shfl_xor(s1[j], 32, 64), o2 = __shfl_xor(s2[j], 32, 64);
        if (half == 0) {
            red[(wave * 2 + 0) * S::HP + 32 * j + l31] = s1[j] + o1;
            red[(wave * 2 + 1) * S::HP + 32 * j + l31] = s2[j] + o2;
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * S::HP) {
        const int k = threadIdx.x / S::HP, n = threadIdx.x % S::HP;
        if (n < H)
            a.partial[((int64_t)blockIdx.x * 2 + k) * H + n] =
                (red[(0 * 2 + k) * S::HP + n] + red[(1 * 2 + k) * S::HP + n]) +
                (red[(2 * 2 + k) * S::HP + n] + red[(3 * 2 + k) * S::HP + n]);
    }
}

template <int NB, int EC>
__global__ __launch_bounds__(256) void k_din_attn2_fwd(DinAttnArgs a) {
    using S = Da2Smem<NB, 4, false, true>;
    __shared__ S sm;
    da2_load_params<S, EC>(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    float* Ps = sm.Ps[wave];
    float* Hq = sm.Hq[wave];
    const int L = a.L, e0 = 8 * half;
    for (int i = lane; i < (32 - 2 * EC) * DA_LDX; i += 64) Xs[2 * EC * DA_LDX + i] = 0.f;
    const float b2 = a.b2 ? a.b2[0] : 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float run = 0.f;
    DaRows cur;
    da_load_rows<EC, false, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        const int c = da2_cut(nvalid, L, lcur);
        const float m = cur.m;
        {
            float hq[2][NB];
            da2_hq<NB, EC>(sm, a, bcur, l31, hq);
            if (half < NB) {
                Hq[32 * half + l31] = hq[0][half < NB ? half : 0];
                Hq[S::HP + 32 * half + l31] = hq[1][half < NB ? half : 0];
            }
        }
        da2_store_x<EC>(Xs, l31, half, cur);
        da_load_rows<EC, false, true>(a, row + 32, row + 32 < R1, half, cur);              // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, true, 2 * EC, S::LDW>(sm.Ws, Xs, 2 * EC, l31, half, acc);
        const float* hqi = Hq + (l31 < c ? 0 : S::HP);
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * j + da_rowmap(r, half);
                const float4 pa = sm.PA[n];
                const float w2 = sm.PB[n].x;
                const float z = acc[j][r] + hqi[n];
                const float zh = (z - pa.y) * pa.z;
                const float p = da_sigmoid(zh);
                const float y = p * z + pa.w * (1.f - p) * z;
                t += y * w2;
            }
        const float o = __shfl_xor(t, 32, 64);
        const float ai = (half == 0 ? t + o : o + t) + b2;
        if (half == 0 && valid) a.a_out[row] = ai;
        const float wm = valid ? ai * m : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u;
            if (e < EC) Ps[e * DA_LDX + l31] = wm * Xs[e * DA_LDX + l31];
        }
        da_seg_sum(Ps, EC, L, lane, nvalid, bcur, lcur, run, a.out, a.out_ld);
    }
}

template <int NB, int EC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_din_attn2_bwd_sums(DinAttnArgs a) {
    using S = Da2Smem<NB, 4, false, false>;
    __shared__ S sm;
    __shared__ float das[4][32];
    da2_load_params<S, EC>(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    const int H = a.H, L = a.L;
    for (int i = lane; i < (32 - 2 * EC) * DA_LDX; i += 64) Xs[2 * EC * DA_LDX + i] = 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float sa[NB], sd[NB], sz[NB], sw[NB];
    float4 pa[NB];
    float w2[NB];
    float sb2 = 0.f;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        sa[j] = 0.f;
        sd[j] = 0.f;
        sz[j] = 0.f;
        sw[j] = 0.f;
        pa[j] = sm.PA[32 * j + l31];
        w2[j] = sm.PB[32 * j + l31].x;
    }
    DaRows cur;
    da_load_rows<EC, true, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        const int c = da2_cut(nvalid, L, lcur);
        float hq[2][NB];
        da2_hq<NB, EC>(sm, a, bcur, l31, hq);
        da2_store_x<EC>(Xs, l31, half, cur);
        float d = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) d = fmaf(cur.dov[u], cur.kv[u], d);
        const float od = __shfl_xor(d, 32, 64);
        const float da_i = (half == 0 ? d + od : od + d) * cur.m;
        if (half == 0) {
            das[wave][l31] = da_i;
            sb2 += da_i;
            if (valid) a.da_out[row] = da_i;
        }
        da_load_rows<EC, true, true>(a, row + 32, row + 32 < R1, half, cur);                // tile t+1
        da_f32x16 acc[NB];
        da_gemm_h<NB, false, 2 * EC, S::LDW>(sm.Ws, Xs, 2 * EC, l31, half, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = da_rowmap(r, half);
            const float dar = das[wave][i];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float z = acc[j][r] + (i < c ? hq[0][j] : hq[1][j]);
                const float zh = (z - pa[j].y) * pa[j].z;
                const float p = da_sigmoid(zh);
                const float y = p * z + pa[j].w * (1.f - p) * z;
                const float dy = dar * w2[j];
                const float dzh = dy * z * (1.f - pa[j].w) * p * (1.f - p);
                sa[j] = fmaf(dy * (1.f - p), z, sa[j]);
                sd[j] += dzh;
                sz[j] = fmaf(dzh, zh, sz[j]);
                sw[j] = fmaf(dar, y, sw[j]);
            }
        }
        da2_advance(nvalid, L, c, bcur, lcur);
    }
    sb2 = fx_wave_sum(sb2);
    __syncthreads();
    float* red = &sm.Xs[0][0];                     // [4 waves][4][HP] + [4] for db2 (4 * 1056 floats)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float oa = __shfl_xor(sa[j], 32, 64), od = __shfl_xor(sd[j], 32, 64);
        const float oz = __shfl_xor(sz[j], 32, 64), ow = __shfl_xor(sw[j], 32, 64);
        if (half == 0) {
            red[(wave * 4 + 0) * S::HP + 32 * j + l31] = sa[j] + oa;
            red[(wave * 4 + 1) * S::HP + 32 * j + l31] = sd[j] + od;
            red[(wave * 4 + 2) * S::HP + 32 * j + l31] = sz[j] + oz;
            red[(wave * 4 + 3) * S::HP + 32 * j + l31] = sw[j] + ow;
        }
    }
    if (lane == 0) red[16 * S::HP + wave] = sb2;
    __syncthreads();
    for (int t = threadIdx.x; t < 4 * S::HP; t += 256) {
        const int k = t / S::HP, n = t % S::HP;
        if (n < H)
            a.partial[((int64_t)blockIdx.x * 5 + k) * H + n] =
                (red[(0 * 4 + k) * S::HP + n] + red[(1 * 4 + k) * S::HP + n]) +
                (red[(2 * 4 + k) * S::HP + n] + red[(3 * 4 + k) * S::HP + n]);
    }
    for (int n = threadIdx.x; n < H; n += 256)
        a.partial[((int64_t)blockIdx.x * 5 + 4) * H + n] =
            n == 0 ? (red[16 * S::HP + 0] + red[16 * S::HP + 1]) +
                         (red[16 * S::HP + 2] + red[16 * S::HP + 3])
                   : 0.f;
}

template <int NB, int EC>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_din_attn2_bwd(DinAttnArgs a) {
    using S = Da2Smem<NB, 2, true, false>;
    __shared__ S sm;
    da2_load_params<S, EC>(sm, a, 128);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    float* Hs = sm.Hs[wave];
    float* Hq = sm.Hq[wave];
    const int H = a.H, L = a.L, KX = 4 * EC;
    const int e0 = 8 * half;
    for (int i = lane; i < (32 - 2 * EC) * DA_LDX; i += 64) Xs[2 * EC * DA_LDX + i] = 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 2 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    da_f32x16 accW[NB];                            // dW'[n][f], f < E: dWb - (c part), f >= E: dWd
    float dWq[NB][EC];                             // unit 32 j + l31 (both halves hold the same)
    float db1q[NB], dhs_run[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) accW[j][r] = 0.f;
#pragma unroll
        for (int e = 0; e < EC; ++e) dWq[j][e] = 0.f;
        db1q[j] = 0.f;
        dhs_run[j] = 0.f;
    }
    uint32_t bcur = (uint32_t)(R0 / L);
    int lcur = 0;
    float dq_run = 0.f;
    DaRows cur;
    da_load_rows<EC, false, true>(a, R0 + l31, R0 + l31 < R1, half, cur);
    float da_n = (R0 + l31 < R1) ? a.da_in[R0 + l31] : 0.f;
    float a_n = (R0 + l31 < R1) ? a.a_in[R0 + l31] : 0.f;
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
        const int c = da2_cut(nvalid, L, lcur);
        const uint32_t b = cur.b;
        const int l = cur.l;
        const float da_i = da_n;
        const float wm = a_n * cur.m;
        float qv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) qv[u] = cur.qv[u];
        {
            float hq[2][NB];
            da2_hq<NB, EC>(sm, a, bcur, l31, hq);
            if (half < NB) {
                Hq[32 * half + l31] = hq[0][half < NB ? half : 0];
                Hq[S::HP + 32 * half + l31] = hq[1][half < NB ? half : 0];
            }
        }
        da2_store_x<EC>(Xs, l31, half, cur);
        da_load_rows<EC, false, true>(a, row + 32, row + 32 < R1, half, cur);              // tile t+1
        da_n = (row + 32 < R1) ? a.da_in[row + 32] : 0.f;
        a_n = (row + 32 < R1) ? a.a_in[row + 32] : 0.f;
        {
            da_f32x16 acc[NB];
            da_gemm_h<NB, true, 2 * EC, S::LDW>(sm.Ws, Xs, 2 * EC, l31, half, acc);
            const float* hqi = Hq + (l31 < c ? 0 : S::HP);
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int n = 32 * j + da_rowmap(r, half);
                    const float4 pa = sm.PA[n];
                    const float4 pb = sm.PB[n];
                    const float z = acc[j][r] + hqi[n];
                    const float zh = (z - pa.y) * pa.z;
                    const float p = da_sigmoid(zh);
                    const float dy = da_i * pb.x;
                    float dzh = dy * z * (1.f - pa.w) * p * (1.f - p);
                    dzh -= pb.y + zh * pb.z;                 // 0 outside training mode
                    float dh = dy * (p + pa.w * (1.f - p)) + dzh * pa.z;
                    if (!valid) dh = 0.f;
                    Hs[l31 * S::LDH + n] = dh;
                }
        }
        // dW'[n][f] += sum_i dh[i][n] x'[i][f]  and  dx'^T[f][i] = sum_n W'[n][f] dh[i][n], interleaved;
        // db1 and the per-sample sums dhs ride along on the A fragments (= dh)
        da_f32x16 accD;
#pragma unroll
        for (int r = 0; r < 16; ++r) accD[r] = 0.f;
        float dhs0[NB], dhs1[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            dhs0[j] = 0.f;
            dhs1[j] = 0.f;
        }
        {
            float av[2][NB], bv[2], bh[2][NB], aw[2][NB];
            auto rd = [&](int kk, int s_) {
#pragma unroll
                for (int j = 0; j < NB; ++j) av[s_][j] = Hs[(2 * kk + half) * S::LDH + 32 * j + l31];
                bv[s_] = Xs[l31 * DA_LDX + 2 * kk + half];
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    const int kd = kk * NB + t;
                    bh[s_][t] = Hs[l31 * S::LDH + 2 * kd + half];
                    aw[s_][t] = sm.Ws[l31 * S::LDW + 2 * kd + half];
                }
            };
            rd(0, 0);
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                const int cs = kk & 1;
                if (kk + 1 < 16) rd(kk + 1, cs ^ 1);
                __builtin_amdgcn_sched_barrier(0);
                const bool first = 2 * kk + half < c;      // this fragment's position: sample bcur?
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    db1q[j] += av[cs][j];
                    if (first) dhs0[j] += av[cs][j];
                    else dhs1[j] += av[cs][j];
                    accW[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][j], bv[cs], accW[j], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t)
                    accD = __builtin_amdgcn_mfma_f32_32x32x2f32(aw[cs][t], bh[cs][t], accD, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        float dov[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) dov[u] = 0.f;
        if (valid && e0 < EC) da_ld8(a.dout + (int64_t)b * a.dout_ld + e0, true, e0, EC, dov);
        // dx'^T through LDS (over the dh tile); rows e: (Wb - Wc)^T dh, rows E + e: Wd^T dh
#pragma unroll
        for (int r = 0; r < 16; ++r) Hs[da_rowmap(r, half) * DA_LDX + l31] = accD[r];
        float dkv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = e0 + u;
            dkv[u] = 0.f;
            if (e < EC) {
                const float ke = Xs[e * DA_LDX + l31];
                const float dxk = Hs[e * DA_LDX + l31];
                const float dxp = Hs[(EC + e) * DA_LDX + l31];
                dkv[u] = (dxk + dxp * qv[u]) + wm * dov[u];
                Xs[(EC + e) * DA_LDX + l31] = valid ? dxp * ke : 0.f;   // share of dq (over q*k rows)
            }
        }
        if (valid && e0 < EC) {
            float* dkp = a.dK + (int64_t)b * a.dk_ldb + (int64_t)l * a.dk_ldl + e0;
            *reinterpret_cast<float4*>(dkp) = make_float4(dkv[0], dkv[1], dkv[2], dkv[3]);
            *reinterpret_cast<float4*>(dkp + 4) = make_float4(dkv[4], dkv[5], dkv[6], dkv[7]);
        }
        // a sample ends inside this tile: its dhs is complete -> dWa += dhs q^T, dq gets (Wa+Wc)^T dhs
        float extra = 0.f;
        if (lcur + c == L) {
            float dfin[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float t = dhs_run[j] + dhs0[j];
                const float o = __shfl_xor(t, 32, 64);
                dfin[j] = half == 0 ? t + o : o + t;
                dhs_run[j] = dhs1[j];
            }
            const float* qp = a.q + (int64_t)bcur * a.q_ld;
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                const float qe = qp[e];
#pragma unroll
                for (int j = 0; j < NB; ++j) dWq[j][e] = fmaf(dfin[j], qe, dWq[j][e]);
            }
            if (half < NB) Hq[32 * half + l31] = dfin[half < NB ? half : 0];   // hq is consumed
            const int e = lane & 15;
            if (e < EC) {
#pragma unroll 8
                for (int n = 0; n < S::HP; ++n) extra = fmaf(sm.Wqs[e * S::LDW + n], Hq[n], extra);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) dhs_run[j] += dhs0[j];
        }
        da_seg_sum(Xs + EC * DA_LDX, EC, L, lane, nvalid, bcur, lcur, dq_run, a.dq, a.dq_ld, extra,
                   a.dq_acc != 0);
    }
    // ---- per-workgroup partial of dW1 (assembled from dWq, dW') and db1
    __syncthreads();
    float* Wt = sm.Hs[wave];                       // this wave's dW' as [n][DA_LDX] (f < 2E)
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Wt[(32 * j + da_rowmap(r, half)) * DA_LDX + l31] = accW[j][r];
    float* Q1 = &sm.Xs[1][0];                      // wave 1's dWq as [n][EC] (HP * EC <= 32 * 33 floats)
    float* bsc = &sm.Hq[0][0];                     // [2 waves][HP] (Hq[0] and Hq[1] are adjacent)
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float o = __shfl_xor(db1q[j], 32, 64);
        if (half == 0) bsc[wave * 2 * S::HP + 32 * j + l31] = db1q[j] + o;
    }
    if (wave == 1 && half < NB) {
#pragma unroll
        for (int e = 0; e < EC; ++e) Q1[(32 * half + l31) * EC + e] = dWq[half < NB ? half : 0][e];
    }
    __syncthreads();
    if (wave == 0 && half < NB) {
        const int n = 32 * half + l31;
        if (n < H) {
            float* P = a.partial + (int64_t)blockIdx.x * ((int64_t)H * KX + H);
            const float* W0 = sm.Hs[0] + n * DA_LDX;
            const float* W1t = sm.Hs[1] + n * DA_LDX;
#pragma unroll
            for (int e = 0; e < EC; ++e) {
                const float qa = dWq[half < NB ? half : 0][e] + Q1[n * EC + e];
                const float kb = W0[e] + W1t[e];
                const float pd = W0[EC + e] + W1t[EC + e];
                P[(int64_t)n * KX + e] = qa;
                P[(int64_t)n * KX + EC + e] = kb;
                P[(int64_t)n * KX + 2 * EC + e] = qa - kb;
                P[(int64_t)n * KX + 3 * EC + e] = pd;
            }
            P[(int64_t)H * KX + n] = bsc[n] + bsc[2 * S::HP + n];
        }
    }
}

// out[k * H + h] = sum over chunks c (fixed order) of partial[(c * nt + k) * H + h]
__global__ __launch_bounds__(256) void k_da_chunks_sum(const float* partial, int chunks, int nt,
                                                       int64_t H, float* out) {
    __shared__ float red[16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int64_t h = (int64_t)blockIdx.x * 16 + tx;
    const int k = blockIdx.y;
    float s = 0.f;
    if (h < H) {
        int c = ty;
        for (; c + 7 * 16 < chunks; c += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[((int64_t)(c + u * 16) * nt + k) * H + h];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; c < chunks; c += 16) s += partial[((int64_t)c * nt + k) * H + h];
    }
    red[ty][tx] = s;
    __syncthreads();
    if (ty == 0 && h < H) {
        float t = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) t += red[y][tx];
        out[(int64_t)k * H + h] = t;
    }
}

__global__ __launch_bounds__(256) void k_da_stats_from_sums(const float* sums, int H, double n_total,
                                                            float momentum, float* stats,
                                                            float* running_mean,
                                                            float* running_var,
                                                            int64_t* num_batches_tracked) {
    const int h = blockIdx.x * 256 + threadIdx.x;
    if (h == 0 && num_batches_tracked) *num_batches_tracked += 1;   // nn.BatchNorm1d's step counter
    if (h >= H) return;
    const double mean = (double)sums[h] / n_total;
    double var = (double)sums[H + h] / n_total - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[h] = (float)mean;
    stats[H + h] = (float)var;
    if (running_mean) {
        const double unb = n_total > 1.0 ? var * n_total / (n_total - 1.0) : var;
        running_mean[h] = (float)((1.0 - momentum) * running_mean[h] + momentum * mean);
        running_var[h] = (float)((1.0 - momentum) * running_var[h] + momentum * unb);
    }
}

// k_da_chunks_sum (k = 0, 1: sum h, sum h^2) and k_da_stats_from_sums in one launch — the single-rank
// case, where nothing (no all-reduce) happens between them.  Same chunk order, same statistics code.
__global__ __launch_bounds__(256) void k_da_chunks_stats(const float* partial, int chunks, int H,
                                                         double n_total, float momentum, float* sums,
                                                         float* stats, float* running_mean,
                                                         float* running_var, int64_t* num_batches_tracked) {
    __shared__ float red[2][16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int h = blockIdx.x * 16 + tx;
    if (blockIdx.x == 0 && threadIdx.x == 0 && num_batches_tracked) *num_batches_tracked += 1;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        float s = 0.f;
        if (h < H) {
            int c = ty;
            for (; c + 7 * 16 < chunks; c += 8 * 16) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = partial[((int64_t)(c + u * 16) * 2 + k) * H + h];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            for (; c < chunks; c += 16) s += partial[((int64_t)c * 2 + k) * H + h];
        }
        red[k][ty][tx] = s;
    }
    __syncthreads();
    if (ty == 0 && h < H) {
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int y = 0; y < 16; ++y) {
            t0 += red[0][y][tx];
            t1 += red[1][y][tx];
        }
        sums[h] = t0;
        sums[H + h] = t1;
        const double mean = (double)t0 / n_total;
        double var = (double)t1 / n_total - mean * mean;
        if (var < 0.0) var = 0.0;
        stats[h] = (float)mean;
        stats[H + h] = (float)var;
        if (running_mean) {
            const double unb = n_total > 1.0 ? var * n_total / (n_total - 1.0) : var;
            running_mean[h] = (float)((1.0 - momentum) * running_mean[h] + momentum * mean);
            running_var[h] = (float)((1.0 - momentum) * running_var[h] + momentum * unb);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DaGeom {
    int64_t n_rows, rpw_flat, wgs_flat, rpw_fwd, wgs_fwd, rpw_bwd, wgs_bwd;
};

static int64_t da_env_cap(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    const int64_t v = e ? atoll(e) : 0;
    return v >= 64 ? v : dflt;
}

static DaGeom da_geom(int64_t B, int32_t L, bool qsplit = false) {
    // waves per launch (experiment switches FX_DIN_ATTN_WAVES / _FWD_WAVES / _BWD_WAVES): a pass is a
    // serial chain per wave (x tile -> MFMA -> gate -> ...), so what matters is that every SIMD gets
    // the same number of tiles (measured: profiles/r02_din_attn_passes.txt)
    static const int64_t cap_flat = da_env_cap("FX_DIN_ATTN_WAVES", 2048);
    static const int64_t cap_fwd = da_env_cap("FX_DIN_ATTN_FWD_WAVES", 2048);
    static const int64_t cap_bwd = da_env_cap("FX_DIN_ATTN_BWD_WAVES", 1536);
    static const int64_t cap_bwd2 = da_env_cap("FX_DIN_ATTN_BWD_WAVES", 2048);   // q split: 4 WGs per CU
    DaGeom g;
    g.n_rows = B * L;
    // statistics passes: 32-position tiles dealt to <= cap_flat waves, no per-sample reduction
    const int64_t nblocks = fx_ceil_div(g.n_rows, 32);
    const int64_t bpw = fx_ceil_div(nblocks, cap_flat) > 1 ? fx_ceil_div(nblocks, cap_flat) : 1;
    g.rpw_flat = bpw * 32;
    g.wgs_flat = fx_ceil_div(fx_ceil_div(nblocks, bpw), 4);
    // apply passes: whole samples per wave (out[b] / dq[b] are finished inside the wave)
    const int64_t Sf = fx_ceil_div(B, cap_fwd) > 1 ? fx_ceil_div(B, cap_fwd) : 1;
    g.rpw_fwd = Sf * L;
    g.wgs_fwd = fx_ceil_div(fx_ceil_div(B, Sf), 4);
    const int64_t cb = qsplit ? cap_bwd2 : cap_bwd;
    const int64_t Sb = fx_ceil_div(B, cb) > 1 ? fx_ceil_div(B, cb) : 1;
    g.rpw_bwd = Sb * L;
    g.wgs_bwd = fx_ceil_div(fx_ceil_div(B, Sb), 2);      // 2 waves per workgroup (LDS)
    if (qsplit) {                                        // every pass owns whole samples
        g.rpw_flat = g.rpw_fwd;
        g.wgs_flat = g.wgs_fwd;
    }
    return g;
}

extern "C" int64_t fx_din_attn_workspace_floats(int64_t B, int32_t L, int32_t E, int32_t H) {
    if (B < 1 || L < 1 || E < 1 || H < 1) return 0;
    int64_t need = 0;
    for (int q = 0; q < 2; ++q) {                       // either formulation may be dispatched
        const DaGeom g = da_geom(B, L, q != 0);
        const int64_t a = g.wgs_flat * 5 * H, b = g.wgs_bwd * ((int64_t)H * 4 * E + H);
        need = a > need ? a : need;
        need = b > need ? b : need;
    }
    return need;
}

static int da_check(const char* who, const float* q, const float* K, int64_t B, int32_t L, int32_t E,
                    int32_t H, const float* W1) {
    FX_CHECK_ARG(B >= 1 && L >= 1 && E >= 1 && E <= 16 && H >= 1 && H <= 64,
                 "%s: bad sizes (1 <= E <= 16, 1 <= H <= 64)", who);
    FX_CHECK_ARG(B * (int64_t)L < ((int64_t)1 << 31), "%s: B * L too large", who);
    FX_CHECK_ARG(q && K && W1, "%s: null pointer", who);
    return FX_OK;
}

static bool da_al16(const void* p, int64_t ld0, int64_t ld1 = 0) {
    return (reinterpret_cast<uintptr_t>(p) & 15) == 0 && ld0 % 4 == 0 && ld1 % 4 == 0;
}

static void da_fill(DinAttnArgs& a, const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                    int64_t k_ldl, int64_t B, int32_t L, int32_t E, int32_t H, const float* W1,
                    const float* b1) {
    memset(&a, 0, sizeof(a));
    a.q = q; a.q_ld = q_ld; a.K = K; a.k_ldb = k_ldb; a.k_ldl = k_ldl;
    a.n_rows = B * L; a.nb = (int32_t)B; a.L = L; a.E = E; a.H = H; a.W1 = W1; a.b1 = b1;
    a.vec = (E % 8 == 0) && da_al16(q, q_ld) && da_al16(K, k_ldb, k_ldl);
}

// (NB, FB, EC): hidden blocks, feature blocks, compile-time E (16 / 8 with 16-byte rows, else 0)
#define DA_LAUNCH(KERNEL, NB_, FB_, EC_, THREADS, GRID, STREAM, ARGS) \
    hipLaunchKernelGGL((KERNEL<NB_, FB_, EC_>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS)
#define DA_DISPATCH_NB(KERNEL, NB_, THREADS, GRID, STREAM, ARGS)                                  \
    do {                                                                                          \
        if (ARGS.vec && ARGS.E == 16) DA_LAUNCH(KERNEL, NB_, 2, 16, THREADS, GRID, STREAM, ARGS); \
        else if (ARGS.vec && ARGS.E == 8) DA_LAUNCH(KERNEL, NB_, 1, 8, THREADS, GRID, STREAM, ARGS); \
        else if (4 * ARGS.E <= 32) DA_LAUNCH(KERNEL, NB_, 1, 0, THREADS, GRID, STREAM, ARGS);     \
        else DA_LAUNCH(KERNEL, NB_, 2, 0, THREADS, GRID, STREAM, ARGS);                           \
    } while (0)
#define DA_DISPATCH(KERNEL, THREADS, GRID, STREAM, ARGS)                                          \
    do {                                                                                          \
        if (ARGS.H <= 32) DA_DISPATCH_NB(KERNEL, 1, THREADS, GRID, STREAM, ARGS);                 \
        else DA_DISPATCH_NB(KERNEL, 2, THREADS, GRID, STREAM, ARGS);                              \
    } while (0)

// the q-split formulation: E = 8 / 16 with 16-byte rows, L >= 32 (FX_DIN_ATTN_QSPLIT=0: never)
static bool da2_ok(const DinAttnArgs& a) {
    static const bool on = []() {
        const char* e = getenv("FX_DIN_ATTN_QSPLIT");
        return !(e && atoi(e) == 0);
    }();
    return on && a.vec && (a.E == 16 || a.E == 8) && a.L >= 32;
}
#define DA2_LAUNCH(KERNEL, THREADS, GRID, STREAM, ARGS)                                            \
    do {                                                                                           \
        if (ARGS.H <= 32 && ARGS.E == 16)                                                          \
            hipLaunchKernelGGL((KERNEL<1, 16>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else if (ARGS.H <= 32)                                                                     \
            hipLaunchKernelGGL((KERNEL<1, 8>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else if (ARGS.E == 16)                                                                     \
            hipLaunchKernelGGL((KERNEL<2, 16>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else                                                                                       \
            hipLaunchKernelGGL((KERNEL<2, 8>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
    } while (0)

extern "C" int fx_din_attn_stats(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                                 int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                                 const float* b1, int32_t H, float* sums, float* workspace,
                                 float* stats, float momentum, float* running_mean, float* running_var,
                                 int64_t* num_batches_tracked, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_stats", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(sums && workspace, "fx_din_attn_stats: null pointer");
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    const bool q2 = da2_ok(a);
    const DaGeom g = da_geom(B, L, q2);
    a.rows_per_wave = g.rpw_flat;
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    if (q2) DA2_LAUNCH(k_din_attn2_stats, 256, g.wgs_flat, s, a);
    else DA_DISPATCH(k_din_attn_stats, 256, g.wgs_flat, s, a);
    if (stats)      // one rank: the sums ARE the batch's — statistics in the same launch
        hipLaunchKernelGGL(k_da_chunks_stats, dim3((unsigned)fx_ceil_div(H, 16)), dim3(256), 0, s,
                           (const float*)workspace, (int)g.wgs_flat, (int)H, (double)(B * (int64_t)L), momentum,
                           sums, stats, running_mean, running_var, num_batches_tracked);
    else
        hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(H, 16), 2), dim3(256), 0, s,
                           (const float*)workspace, (int)g.wgs_flat, 2, (int64_t)H, sums);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_stats_from_sums(const float* sums, int32_t H, int64_t n_total, float momentum,
                                       int32_t training, float* running_mean, float* running_var,
                                       int64_t* num_batches_tracked, float* stats, fx_stream_t stream) {
    FX_CHECK_ARG(H >= 1 && stats, "fx_dice_stats_from_sums: bad arguments");
    hipStream_t s = fx_hip_stream(stream);
    if (training) {
        FX_CHECK_ARG(sums && n_total >= 1, "fx_dice_stats_from_sums: training mode needs the sums");
        hipLaunchKernelGGL(k_da_stats_from_sums, dim3((unsigned)fx_ceil_div(H, 256)), dim3(256), 0, s,
                           sums, (int)H, (double)n_total, momentum, stats, running_mean, running_var,
                           num_batches_tracked);
        FX_CHECK_LAUNCH();
    } else {
        FX_CHECK_ARG(running_mean && running_var, "fx_dice_stats_from_sums: null running statistics");
        FX_CHECK_HIP(hipMemcpyAsync(stats, running_mean, sizeof(float) * H, hipMemcpyDeviceToDevice, s));
        FX_CHECK_HIP(hipMemcpyAsync(stats + H, running_var, sizeof(float) * H,
                                    hipMemcpyDeviceToDevice, s));
    }
    return FX_OK;
}

extern "C" int fx_din_attn_fwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                               int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                               const float* b1, int32_t H, const float* alpha, float eps,
                               const float* stats, const float* W2, const float* b2,
                               const int32_t* mask, int64_t mask_ld, float* a_out, float* out,
                               int64_t out_ld, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_fwd", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && a_out && out, "fx_din_attn_fwd: null pointer");
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    const bool q2 = da2_ok(a);
    const DaGeom g = da_geom(B, L, q2);
    a.rows_per_wave = g.rpw_fwd;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2; a.b2 = b2;
    a.mask = mask; a.m_ld = mask_ld; a.a_out = a_out; a.out = out; a.out_ld = out_ld;
    hipStream_t s = fx_hip_stream(stream);
    if (q2) DA2_LAUNCH(k_din_attn2_fwd, 256, g.wgs_fwd, s, a);
    else DA_DISPATCH(k_din_attn_fwd, 256, g.wgs_fwd, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_attn_bwd_sums(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                                    int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                                    const float* b1, int32_t H, const float* alpha, float eps,
                                    const float* stats, const float* W2, const int32_t* mask,
                                    int64_t mask_ld, const float* dout, int64_t dout_ld, float* da,
                                    float* sums5, float* workspace, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_bwd_sums", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && dout && da && sums5 && workspace,
                 "fx_din_attn_bwd_sums: null pointer");
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.vec = a.vec && da_al16(dout, dout_ld);
    const bool q2 = da2_ok(a);
    const DaGeom g = da_geom(B, L, q2);
    a.rows_per_wave = g.rpw_flat;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2;
    a.mask = mask; a.m_ld = mask_ld; a.dout = dout; a.dout_ld = dout_ld; a.da_out = da;
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    if (q2) DA2_LAUNCH(k_din_attn2_bwd_sums, 256, g.wgs_flat, s, a);
    else DA_DISPATCH(k_din_attn_bwd_sums, 256, g.wgs_flat, s, a);
    hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(H, 16), 5), dim3(256), 0, s,
                       (const float*)workspace, (int)g.wgs_flat, 5, (int64_t)H, sums5);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_attn_bwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                               int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                               const float* b1, int32_t H, const float* alpha, float eps,
                               int32_t training, const float* stats, const float* W2,
                               const int32_t* mask, int64_t mask_ld, const float* a_logit,
                               const float* dout, int64_t dout_ld, const float* da,
                               const float* sums5, int64_t n_total, float* dq, int64_t dq_ld,
                               int32_t dq_accumulate, float* dK, int64_t dk_ldb, int64_t dk_ldl,
                               float* dW1b1, float* workspace, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_bwd", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && a_logit && dout && da && dq && dK && dW1b1 && workspace,
                 "fx_din_attn_bwd: null pointer");
    FX_CHECK_ARG(!training || (sums5 && n_total >= B * (int64_t)L),
                 "fx_din_attn_bwd: training mode needs the backward sums and the global row count");
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.vec = a.vec && da_al16(dout, dout_ld) && da_al16(dK, dk_ldb, dk_ldl);
    const bool q2 = da2_ok(a);
    const DaGeom g = da_geom(B, L, q2);
    a.rows_per_wave = g.rpw_bwd;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2;
    a.mask = mask; a.m_ld = mask_ld; a.a_in = a_logit; a.dout = dout; a.dout_ld = dout_ld;
    a.da_in = da;
    a.sums = training ? sums5 : nullptr;
    a.inv_n = training ? 1.f / (float)n_total : 0.f;
    a.dq = dq; a.dq_ld = dq_ld; a.dq_acc = dq_accumulate; a.dK = dK; a.dk_ldb = dk_ldb; a.dk_ldl = dk_ldl;
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    if (q2) DA2_LAUNCH(k_din_attn2_bwd, 128, g.wgs_bwd, s, a);
    else DA_DISPATCH(k_din_attn_bwd, 128, g.wgs_bwd, s, a);
    const int64_t tot = (int64_t)H * 4 * E + H;
    hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(tot, 16), 1), dim3(256), 0, s,
                       (const float*)workspace, (int)g.wgs_bwd, 1, tot, dW1b1);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
