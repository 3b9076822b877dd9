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
//             all-reduce across ranks] -> apply pass (h -> Dice -> . W2 -> a[B*L])
//   backward: sums pass   (h, da -> dalpha, sum dzhat, sum dzhat*zhat, dW2, db2)  -> [all-reduce]
//             -> apply pass (h -> dh -> dW1 / db1 partials, dx -> dq, dK)
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
    int64_t rows_per_wave;   // contiguous positions a wave owns (bwd apply: whole samples)
    int32_t L, E, H;
    int32_t vec;             // q / K / dK rows may be moved with 16-byte accesses (E % 8 == 0, aligned)
    const float* W1;         // [H, 4E]
    const float* b1;         // [H] or null
    const float* alpha;      // [H]
    const float* stats;      // mean[H], biased var[H]
    float eps;
    const float* W2;         // [H]
    const float* b2;         // [1] or null
    const float* da;         // [B*L] gradient of the attention logits
    const float* sums;       // [H + n] = sum dzhat, [2H + n] = sum dzhat * zhat (training mode)
    float inv_n;
    const float* dk_add;     // optional [B, L, E] addend of dK (the pooling's share)
    int64_t dka_ldb, dka_ldl;
    float* a_out;            // [B*L]
    float* dq;
    int64_t dq_ld;
    float* dK;
    int64_t dk_ldb, dk_ldl;
    float* partial;          // per-workgroup partial sums
};

__device__ __forceinline__ int da_rowmap(int r, int half) { return (r & 3) + 8 * (r >> 2) + 4 * half; }

template <int NB, int FB, int WAVES, bool BWD>
struct DaSmem {
    static constexpr int HP = 32 * NB, FP = 32 * FB, LDW = HP + 1, LDH = HP + 1;
    float W1s[FP * LDW];                        // W1^T, [feature][hidden]: W1s[f * LDW + n] = W1[n][f]
    float4 PA[HP];                              // {b1, mean, rstd, alpha}
    float4 PB[HP];                              // {w2, mean(dzhat), mean(dzhat zhat), 0}
    float Xs[WAVES][FP * DA_LDX];               // x^T tile per wave (backward: reused for dx^T)
    static constexpr int HSZ = 32 * LDH > FP * DA_LDX ? 32 * LDH : FP * DA_LDX;
    float Hs[BWD ? WAVES : 1][BWD ? HSZ : 1];   // dh tile per wave, [position][hidden]; then dx^T
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

// The q / K values of one position as the lanes hold them: lane (l31 = position, half) owns features
// [8 half, 8 half + 8).  Loading (global) and writing the x^T tile (LDS) are separate steps so that the
// rows of tile t+1 are requested before the matrix products of tile t (the only HBM latency of a pass).
struct DaRows {
    float qv[8], kv[8];
    int64_t b;
    int l;
};

__device__ __forceinline__ void da_load_rows(const DinAttnArgs& a, int64_t row, bool valid, int half,
                                             DaRows& x) {
    x.b = 0;
    x.l = 0;
    if (valid) {
        x.b = row / a.L;
        x.l = (int)(row - x.b * a.L);
    }
    const int E = a.E, e0 = 8 * half;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        x.qv[u] = 0.f;
        x.kv[u] = 0.f;
    }
    if (valid && e0 < E) {
        const float* qp = a.q + x.b * a.q_ld + e0;
        const float* kp = a.K + x.b * a.k_ldb + (int64_t)x.l * a.k_ldl + e0;
        if (a.vec) {
            const float4 q0 = *reinterpret_cast<const float4*>(qp);
            const float4 q1 = *reinterpret_cast<const float4*>(qp + 4);
            const float4 k0 = *reinterpret_cast<const float4*>(kp);
            const float4 k1 = *reinterpret_cast<const float4*>(kp + 4);
            x.qv[0] = q0.x; x.qv[1] = q0.y; x.qv[2] = q0.z; x.qv[3] = q0.w;
            x.qv[4] = q1.x; x.qv[5] = q1.y; x.qv[6] = q1.z; x.qv[7] = q1.w;
            x.kv[0] = k0.x; x.kv[1] = k0.y; x.kv[2] = k0.z; x.kv[3] = k0.w;
            x.kv[4] = k1.x; x.kv[5] = k1.y; x.kv[6] = k1.z; x.kv[7] = k1.w;
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (e0 + u < E) {
                    x.qv[u] = qp[u];
                    x.kv[u] = kp[u];
                }
            }
        }
    }
}

// x^T[f][position] = [q, k, q - k, q * k] of the tile (rows outside the range were loaded as 0)
__device__ __forceinline__ void da_store_x(float* Xs, int E, int l31, int half, const DaRows& x) {
    const int e0 = 8 * half;
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
// Same LDS reads either way; only the operand order of the MFMA changes.
template <int NB, bool POS_IN_LANE>
__device__ __forceinline__ void da_gemm_h(const float* __restrict__ W1s, int ldw,
                                          const float* __restrict__ Xs, int KX, int l31, int half,
                                          da_f32x16 (&acc)[NB]) {
#pragma unroll
    for (int j = 0; j < NB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float* xb = Xs + half * DA_LDX + l31;
    const float* wa = W1s + half * ldw + l31;
    const int nk = KX >> 1;
#pragma unroll 4
    for (int kk = 0; kk < nk; ++kk) {
        const float bx = xb[2 * kk * DA_LDX];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float aw = wa[2 * kk * ldw + 32 * j];
            if constexpr (POS_IN_LANE)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bx, acc[j], 0, 0, 0);
            else
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(bx, aw, acc[j], 0, 0, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward, pass 1: per-workgroup partial sums of h and h^2 over the positions
//   partial[(wg * 2 + k) * H + n]
// ---------------------------------------------------------------------------------------------
template <int NB, int FB>
__global__ __launch_bounds__(256) void k_din_attn_stats(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false>;
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
    da_load_rows(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        da_store_x(Xs, a.E, l31, half, cur);
        da_load_rows(a, rb + 32 + l31, rb + 32 + l31 < R1, half, cur);     // tile t+1 in flight
        da_f32x16 acc[NB];
        da_gemm_h<NB, false>(sm.W1s, S::LDW, Xs, KX, l31, half, acc);
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
// forward, pass 2: a[b*L + l] = W2 . Dice(W1 x + b1) + b2
// ---------------------------------------------------------------------------------------------
template <int NB, int FB>
__global__ __launch_bounds__(256) void k_din_attn_fwd(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false>;
    __shared__ S sm;
    da_load_params(sm, a, 256);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    const int KX = 4 * a.E;
    const float b2 = a.b2 ? a.b2[0] : 0.f;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t R0 = gw * a.rows_per_wave;
    const int64_t R1 = (R0 + a.rows_per_wave < a.n_rows) ? R0 + a.rows_per_wave : a.n_rows;
    DaRows cur;
    da_load_rows(a, R0 + l31, R0 + l31 < R1, half, cur);
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        da_store_x(Xs, a.E, l31, half, cur);
        da_load_rows(a, row + 32, row + 32 < R1, half, cur);               // tile t+1 in flight
        da_f32x16 acc[NB];
        da_gemm_h<NB, true>(sm.W1s, S::LDW, Xs, KX, l31, half, acc);
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
        const float s = half == 0 ? t + o : o + t;        // units of half 0 first, in both lanes
        if (half == 0 && valid) a.a_out[row] = s + b2;
    }
}

// ---------------------------------------------------------------------------------------------
// backward, pass 1: per-workgroup partial sums over the positions
//   k = 0: dalpha[n]  1: sum dzhat[n]  2: sum dzhat zhat[n]  3: dW2[n]  4: db2 (at n = 0)
//   partial[(wg * 5 + k) * H + n]
// ---------------------------------------------------------------------------------------------
template <int NB, int FB>
__global__ __launch_bounds__(256) void k_din_attn_bwd_sums(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 4, false>;
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
    da_load_rows(a, R0 + l31, R0 + l31 < R1, half, cur);
    float da_i = (R0 + l31 < R1) ? a.da[R0 + l31] : 0.f;
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        da_store_x(Xs, a.E, l31, half, cur);
        if (half == 0) {
            das[wave][l31] = da_i;
            sb2 += da_i;
        }
        da_load_rows(a, row + 32, row + 32 < R1, half, cur);               // tile t+1 in flight
        da_i = (row + 32 < R1) ? a.da[row + 32] : 0.f;
        da_f32x16 acc[NB];
        da_gemm_h<NB, false>(sm.W1s, S::LDW, Xs, KX, l31, half, acc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float dar = das[wave][da_rowmap(r, half)];     // 0 for positions past the range
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
// backward, pass 2: dh -> dW1 / db1 partials (per workgroup), dq, dK
//   partial[wg * (H * 4E + H) + n * 4E + f]  and  [... + H * 4E + n]
// A wave owns whole samples (rows_per_wave is a multiple of L), so dq[b] is finished by one lane per
// feature walking the positions in order.
// ---------------------------------------------------------------------------------------------
template <int NB, int FB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(2, 2)))
void k_din_attn_bwd(DinAttnArgs a) {
    using S = DaSmem<NB, FB, 2, true>;
    __shared__ S sm;
    da_load_params(sm, a, 128);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    float* Xs = sm.Xs[wave];
    float* Hs = sm.Hs[wave];
    const int E = a.E, KX = 4 * E, H = a.H, L = a.L;
    const int e0 = 8 * half;
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
    float dq_run = 0.f;
    DaRows cur;
    da_load_rows(a, R0 + l31, R0 + l31 < R1, half, cur);
    float da_n = (R0 + l31 < R1) ? a.da[R0 + l31] : 0.f;
    for (int64_t rb = R0; rb < R1; rb += 32) {
        const int64_t row = rb + l31;
        const bool valid = row < R1;
        const int64_t b = cur.b;
        const int l = cur.l;
        const float da_i = da_n;
        da_store_x(Xs, E, l31, half, cur);
        da_load_rows(a, row + 32, row + 32 < R1, half, cur);               // tile t+1 in flight
        da_n = (row + 32 < R1) ? a.da[row + 32] : 0.f;
        {
            da_f32x16 acc[NB];
            da_gemm_h<NB, true>(sm.W1s, S::LDW, Xs, KX, l31, half, acc);
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
        // dW1[n][f] += sum_i dh[i][n] x[i][f]
#pragma unroll 2
        for (int kk = 0; kk < 16; ++kk) {
            float av[NB], bv[FB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                av[j] = Hs[(2 * kk + half) * S::LDH + 32 * j + l31];
                db1q[j] += av[j];                  // db1 rides along: the A fragments ARE dh
            }
#pragma unroll
            for (int jb = 0; jb < FB; ++jb) bv[jb] = Xs[(32 * jb + l31) * DA_LDX + 2 * kk + half];
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int jb = 0; jb < FB; ++jb)
                    accW[j][jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[jb], accW[j][jb],
                                                                       0, 0, 0);
        }
        // dx^T[f][i] = sum_n W1[n][f] dh[i][n]
        da_f32x16 accD[FB];
#pragma unroll
        for (int jb = 0; jb < FB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r) accD[jb][r] = 0.f;
#pragma unroll 4
        for (int kk = 0; kk < S::HP / 2; ++kk) {
            const float bh = Hs[l31 * S::LDH + 2 * kk + half];
#pragma unroll
            for (int jb = 0; jb < FB; ++jb) {
                const float aw = sm.W1s[(32 * jb + l31) * S::LDW + 2 * kk + half];
                accD[jb] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw, bh, accD[jb], 0, 0, 0);
            }
        }
        // dx^T through LDS, over the dh tile (both products have consumed it); the x tile still holds
        // this tile's q and k rows
#pragma unroll
        for (int jb = 0; jb < FB; ++jb)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Hs[(32 * jb + da_rowmap(r, half)) * DA_LDX + l31] = accD[jb][r];
        // dk = dx_k - dx_(q-k) + dx_(q*k) q ;  this position's share of dq = dx_q + dx_(q-k) + dx_(q*k) k
        // (written over the (q-k) rows of the x tile, which nobody reads any more)
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
                dkv[u] = dxb - dxc + dxd * qe;
                const float dqc = dxa + dxc + dxd * ke;
                Xs[(2 * E + e) * DA_LDX + l31] = valid ? dqc : 0.f;
            }
        }
        if (valid && e0 < E) {
            float* dkp = a.dK + b * a.dk_ldb + (int64_t)l * a.dk_ldl + e0;
            if (a.dk_add) {
                const float* ap = a.dk_add + b * a.dka_ldb + (int64_t)l * a.dka_ldl + e0;
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < E) dkv[u] += ap[u];
            }
            if (a.vec) {
                *reinterpret_cast<float4*>(dkp) = make_float4(dkv[0], dkv[1], dkv[2], dkv[3]);
                *reinterpret_cast<float4*>(dkp + 4) = make_float4(dkv[4], dkv[5], dkv[6], dkv[7]);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (e0 + u < E) dkp[u] = dkv[u];
            }
        }
        // dq: lane e walks the tile's positions in order; a sample ends after its L-th position
        {
            int64_t bcur = rb / L;
            int lcur = (int)(rb - bcur * L);
            const int nvalid = (R1 - rb < 32) ? (int)(R1 - rb) : 32;
            for (int i = 0; i < nvalid; ++i) {
                if (lane < E) dq_run += Xs[(2 * E + lane) * DA_LDX + i];
                if (++lcur == L) {
                    if (lane < E) a.dq[bcur * a.dq_ld + lane] = dq_run;
                    dq_run = 0.f;
                    lcur = 0;
                    ++bcur;
                }
            }
        }
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
                                                            float* running_var) {
    const int h = blockIdx.x * 256 + threadIdx.x;
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

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct DaGeom {
    int64_t n_rows, rpw13, wgs13, rpw4, wgs4;
};

static int64_t da_env_cap(const char* name, int64_t dflt) {
    const char* e = getenv(name);
    const int64_t v = e ? atoll(e) : 0;
    return v >= 64 ? v : dflt;
}

static DaGeom da_geom(int64_t B, int32_t L) {
    // waves per launch (experiment switches FX_DIN_ATTN_WAVES / FX_DIN_ATTN_BWD_WAVES): the passes are
    // a serial chain per wave (x tile -> MFMA -> gate -> ...), so what matters is that every SIMD gets
    // the same number of tiles, not how many waves are resident
    static const int64_t cap13 = da_env_cap("FX_DIN_ATTN_WAVES", 2048);
    static const int64_t cap4 = da_env_cap("FX_DIN_ATTN_BWD_WAVES", 1536);
    DaGeom g;
    g.n_rows = B * L;
    // passes without a per-sample reduction: 32-position tiles dealt to <= cap13 waves
    const int64_t nblocks = fx_ceil_div(g.n_rows, 32);
    const int64_t bpw = fx_ceil_div(nblocks, cap13) > 1 ? fx_ceil_div(nblocks, cap13) : 1;
    g.rpw13 = bpw * 32;
    g.wgs13 = fx_ceil_div(fx_ceil_div(nblocks, bpw), 4);
    // backward apply: whole samples per wave, <= cap4 waves (3 workgroups of 2 waves per CU: LDS)
    const int64_t S = fx_ceil_div(B, cap4) > 1 ? fx_ceil_div(B, cap4) : 1;
    g.rpw4 = S * L;
    g.wgs4 = fx_ceil_div(fx_ceil_div(B, S), 2);
    return g;
}

extern "C" int64_t fx_din_attn_workspace_floats(int64_t B, int32_t L, int32_t E, int32_t H) {
    if (B < 1 || L < 1 || E < 1 || H < 1) return 0;
    const DaGeom g = da_geom(B, L);
    const int64_t a = g.wgs13 * 5 * H, b = g.wgs4 * ((int64_t)H * 4 * E + H);
    return a > b ? a : b;
}

static int da_check(const char* who, const float* q, const float* K, int64_t B, int32_t L, int32_t E,
                    int32_t H, const float* W1) {
    FX_CHECK_ARG(B >= 1 && L >= 1 && E >= 1 && E <= 16 && H >= 1 && H <= 64,
                 "%s: bad sizes (1 <= E <= 16, 1 <= H <= 64)", who);
    FX_CHECK_ARG(B * (int64_t)L < ((int64_t)1 << 31), "%s: B * L too large", who);
    FX_CHECK_ARG(q && K && W1, "%s: null pointer", who);
    return FX_OK;
}

static void da_fill(DinAttnArgs& a, const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                    int64_t k_ldl, int64_t B, int32_t L, int32_t E, int32_t H, const float* W1,
                    const float* b1) {
    memset(&a, 0, sizeof(a));
    a.q = q; a.q_ld = q_ld; a.K = K; a.k_ldb = k_ldb; a.k_ldl = k_ldl;
    a.n_rows = B * L; a.L = L; a.E = E; a.H = H; a.W1 = W1; a.b1 = b1;
    a.vec = (E % 8 == 0) && (q_ld % 4 == 0) && (k_ldb % 4 == 0) && (k_ldl % 4 == 0) &&
            (((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(K)) & 15) == 0);
}

#define DA_DISPATCH(KERNEL, THREADS, GRID, STREAM, ARGS)                                             \
    do {                                                                                             \
        const int nb_ = (ARGS.H + 31) / 32, fb_ = (4 * ARGS.E + 31) / 32;                            \
        if (nb_ == 1 && fb_ == 1)                                                                    \
            hipLaunchKernelGGL((KERNEL<1, 1>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else if (nb_ == 1)                                                                           \
            hipLaunchKernelGGL((KERNEL<1, 2>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else if (fb_ == 1)                                                                           \
            hipLaunchKernelGGL((KERNEL<2, 1>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
        else                                                                                         \
            hipLaunchKernelGGL((KERNEL<2, 2>), dim3((unsigned)(GRID)), dim3(THREADS), 0, STREAM, ARGS); \
    } while (0)

extern "C" int fx_din_attn_stats(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                                 int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                                 const float* b1, int32_t H, float* sums, float* workspace,
                                 fx_stream_t stream) {
    int rc = da_check("fx_din_attn_stats", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(sums && workspace, "fx_din_attn_stats: null pointer");
    const DaGeom g = da_geom(B, L);
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.rows_per_wave = g.rpw13;
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    DA_DISPATCH(k_din_attn_stats, 256, g.wgs13, s, a);
    hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(H, 16), 2), dim3(256), 0, s,
                       (const float*)workspace, (int)g.wgs13, 2, (int64_t)H, sums);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dice_stats_from_sums(const float* sums, int32_t H, int64_t n_total, float momentum,
                                       int32_t training, float* running_mean, float* running_var,
                                       float* stats, fx_stream_t stream) {
    FX_CHECK_ARG(H >= 1 && stats, "fx_dice_stats_from_sums: bad arguments");
    hipStream_t s = fx_hip_stream(stream);
    if (training) {
        FX_CHECK_ARG(sums && n_total >= 1, "fx_dice_stats_from_sums: training mode needs the sums");
        hipLaunchKernelGGL(k_da_stats_from_sums, dim3((unsigned)fx_ceil_div(H, 256)), dim3(256), 0, s,
                           sums, (int)H, (double)n_total, momentum, stats, running_mean, running_var);
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
                               const float* stats, const float* W2, const float* b2, float* a_out,
                               fx_stream_t stream) {
    int rc = da_check("fx_din_attn_fwd", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && a_out, "fx_din_attn_fwd: null pointer");
    const DaGeom g = da_geom(B, L);
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.rows_per_wave = g.rpw13;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2; a.b2 = b2; a.a_out = a_out;
    hipStream_t s = fx_hip_stream(stream);
    DA_DISPATCH(k_din_attn_fwd, 256, g.wgs13, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_attn_bwd_sums(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                                    int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                                    const float* b1, int32_t H, const float* alpha, float eps,
                                    const float* stats, const float* W2, const float* da,
                                    float* sums5, float* workspace, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_bwd_sums", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && da && sums5 && workspace,
                 "fx_din_attn_bwd_sums: null pointer");
    const DaGeom g = da_geom(B, L);
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.rows_per_wave = g.rpw13;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2; a.da = da;
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    DA_DISPATCH(k_din_attn_bwd_sums, 256, g.wgs13, s, a);
    hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(H, 16), 5), dim3(256), 0, s,
                       (const float*)workspace, (int)g.wgs13, 5, (int64_t)H, sums5);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_din_attn_bwd(const float* q, int64_t q_ld, const float* K, int64_t k_ldb,
                               int64_t k_ldl, int64_t B, int32_t L, int32_t E, const float* W1,
                               const float* b1, int32_t H, const float* alpha, float eps,
                               int32_t training, const float* stats, const float* W2,
                               const float* da, const float* sums5, int64_t n_total,
                               const float* dk_add, int64_t dka_ldb, int64_t dka_ldl, float* dq,
                               int64_t dq_ld, float* dK, int64_t dk_ldb, int64_t dk_ldl,
                               float* dW1b1, float* workspace, fx_stream_t stream) {
    int rc = da_check("fx_din_attn_bwd", q, K, B, L, E, H, W1);
    if (rc != FX_OK) return rc;
    FX_CHECK_ARG(alpha && stats && W2 && da && dq && dK && dW1b1 && workspace,
                 "fx_din_attn_bwd: null pointer");
    FX_CHECK_ARG(!training || (sums5 && n_total >= B * (int64_t)L),
                 "fx_din_attn_bwd: training mode needs the backward sums and the global row count");
    const DaGeom g = da_geom(B, L);
    DinAttnArgs a;
    da_fill(a, q, q_ld, K, k_ldb, k_ldl, B, L, E, H, W1, b1);
    a.rows_per_wave = g.rpw4;
    a.alpha = alpha; a.eps = eps; a.stats = stats; a.W2 = W2; a.da = da;
    a.sums = training ? sums5 : nullptr;
    a.inv_n = training ? 1.f / (float)n_total : 0.f;
    a.dk_add = dk_add; a.dka_ldb = dka_ldb; a.dka_ldl = dka_ldl;
    a.dq = dq; a.dq_ld = dq_ld; a.dK = dK; a.dk_ldb = dk_ldb; a.dk_ldl = dk_ldl;
    a.vec = a.vec && (dk_ldb % 4 == 0) && (dk_ldl % 4 == 0) &&
            ((reinterpret_cast<uintptr_t>(dK) & 15) == 0);
    a.partial = workspace;
    hipStream_t s = fx_hip_stream(stream);
    DA_DISPATCH(k_din_attn_bwd, 128, g.wgs4, s, a);
    const int64_t tot = (int64_t)H * 4 * E + H;
    hipLaunchKernelGGL(k_da_chunks_sum, dim3((unsigned)fx_ceil_div(tot, 16), 1), dim3(256), 0, s,
                       (const float*)workspace, (int)g.wgs4, 1, tot, dW1b1);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
