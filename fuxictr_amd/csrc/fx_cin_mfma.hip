// fx_cin_mfma.hip — the CIN layer of xDeepFM on the matrix cores.
// Reference: fuxictr/pytorch/layers/interactions/compressed_interaction_net.py:54-76
#include "fx_cin.h"

#include <stdlib.h>

// =================================================================================================
// Round 3: the CIN products on the matrix cores (v_mfma_f32_16x16x4_f32: exact fp32 products, fp32
// accumulate, the same 64 flop/clk/SIMD peak as the 32x32x2 form) for the shape class of the BASELINE
// configuration: D = 16 (the 16 dims of ONE sample are the 16 columns of an MFMA tile), O <= 16 (rows),
// F0, Mi <= 40.  The compress step of compressed_interaction_net.py:70-74,
//     Xn[o, d] = sum_{h,m} W[o, (h,m)] X0[h,d] Xi[m,d]  =  sum_h X0[h,d] * ( sum_m W[o,(h,m)] Xi[m,d] ),
// is, per sample and per row h of X0, a [16 x Mi] x [Mi x 16] matrix product whose B operand is Xi
// itself; the factor X0[h,d] multiplies the finished 16x16 tile (4 fmac per lane).  Measured on the
// part (scripts/ubench/mfma_rate.hip): an MFMA whose operand comes out of a VALU multiply issues every
// 53 cycles from one wave, a bare one every 33 — so no kernel here forms the outer product
// X0[h,d] * Xi[m,d] in front of the matrix pipe, and every workgroup runs 8 waves (2 per SIMD) so one
// wave's tile arithmetic overlaps the other's MFMAs.  A wave carries NS samples through the K loop.
//   lane l: r = l & 15 (row of A / column of B and D), kk = l >> 4 (k of A and B; D holds rows 4*kk + i)
//   forward   T_h = W_h Xi:   A = W[o = r][(h, m = 4*mq + kk)]       B = Xi[4*mq + kk][d = r]
//             Xn[o = 4*kk + i][d = r] += T_h[i] * X0[h][r]
//   backward  T[(h,m), d] = sum_o W[o,(h,m)] g[o,d]:  A = W^T[m = 16*mt + r][o = 4*j + kk], B = g[4*j + kk][r]
//             dX0[h,d] = sum_m T Xi[m,d] (registers + two shuffles), dXi[m,d] = sum_h T X0[h,d] (registers)
//   weights   dW[o,(h,m)] = sum_{b,d} (g[b,o,d] X0[b,h,d]) Xi[b,m,d]:  A = g[o = r][d = 4*j + kk] * X0[h][4*j + kk]
//             (4 products per h, shared by the MT column tiles and formed a whole h-sweep ahead of their
//             MFMAs), B = Xi[16*mt + r][4*j + kk]; a workgroup's eight waves split the rows h and keep
//             their dW tiles in accumulators over all its samples -> partial[G] as before.
// fp32 VALU kernels above stay for every other shape.
// =================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define FX_CIN_MFMA_NS 2      // samples per wave
#define FX_CIN_WAVES 8        // waves per workgroup (512 threads, 2 per SIMD)

static int fx_cin_mfma_mode() {     // FX_CIN_MFMA=0: the VALU kernels everywhere (A/B runs)
    static const int mode = []() {
        const char* e = getenv("FX_CIN_MFMA");
        return e ? atoi(e) : 1;
    }();
    return mode;
}

#define FX_CIN_HB 5      // rows of X0 fetched per register block (two blocks in flight)

// the two instantiations: Mi <= 16 (layers fed by a 16-map layer) and Mi <= 40 (layer 1 of 39/40 fields)
__host__ __device__ __forceinline__ int fx_cin_mq(int Mi) { return Mi <= 16 ? 4 : 10; }
__host__ __device__ __forceinline__ int fx_cin_mt(int Mi) { return Mi <= 16 ? 1 : 3; }
__host__ __device__ __forceinline__ int fx_cin_rows_padded(int F0) {       // whole blocks of 2*HB rows
    return (F0 + 2 * FX_CIN_HB - 1) / (2 * FX_CIN_HB) * (2 * FX_CIN_HB);
}

// The LDS images of W, laid out so that a wave's A fragment is one conflict-free ds_read_b32:
//   forward  [q = h*MQ + mq][lane]        = W[o = r][h*Mi + 4*mq + kk]          MQ = ceil(Mi / 4)
//   dX       [tile = h*MT + mt][j][lane]  = W[o = 4*j + kk][h*Mi + 16*mt + r]   MT = ceil(Mi / 16)
// zero where o >= O or m >= Mi.  fx_cin_pack_w writes both once per step into w_img (forward image
// first), and every workgroup copies its image with float4 loads; without w_img a workgroup gathers
// the image from W itself (same values, ~100 dependent-latency loads per thread: the slow start).
__device__ __forceinline__ float fx_cin_wq_elem(const float* W, int F0, int Mi, int O, int MQ, int e) {
    const int q = e >> 6, l = e & 63;
    const int o = l & 15, m = (q % MQ) * 4 + (l >> 4), h = q / MQ;
    return (o < O && m < Mi) ? W[(int64_t)o * F0 * Mi + h * Mi + m] : 0.f;
}

__device__ __forceinline__ float fx_cin_wb_elem(const float* W, int F0, int Mi, int O, int MT, int e) {
    const int tile = e >> 8, j = (e >> 6) & 3, l = e & 63;
    const int o = 4 * j + (l >> 4), m = (tile % MT) * 16 + (l & 15), h = tile / MT;
    return (o < O && m < Mi) ? W[(int64_t)o * F0 * Mi + h * Mi + m] : 0.f;
}

__global__ __launch_bounds__(256) void k_cin_pack_w(CinPackArgs pa) {      // blockIdx.y = layer
    const CinPackArgs* ka = (const CinPackArgs*)__builtin_amdgcn_kernarg_segment_ptr();
    const int L = blockIdx.y;
    const float* W = ka->W[L];
    float* img = ka->img[L];
    const int F0 = ka->F0[L], Mi = ka->Mi[L], O = ka->O[L];
    const int MQ = fx_cin_mq(Mi), MT = fx_cin_mt(Mi);
    const int nf = F0 * MQ * 64, nd = F0 * MT * 256;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < nf + nd; e += gridDim.x * 256)
        img[e] = e < nf ? fx_cin_wq_elem(W, F0, Mi, O, MQ, e) : fx_cin_wb_elem(W, F0, Mi, O, MT, e - nf);
}

// Sample rows (X0, Xi, dX0, dXi) are accessed through buffer resources: the address is one VGPR offset
// per sample and block + an immediate, no per-access 64-bit arithmetic, and a read past the tensor's
// last valid float returns 0.  Rows past F0 / Mi of any other sample read its neighbour's (finite) data
// and meet the zero rows / columns of the W image.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t fx_cin_rsrc(const float* p, int64_t B, int64_t ld, int rows) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p), 0,
                                             (int)(((B - 1) * ld + (int64_t)rows * 16) * 4), 0x00020000);
}

// the whole byte offset travels in the VGPR + immediate (the part the range check sees)
__device__ __forceinline__ float fx_cin_bload(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff, 0, 0));
}

#define FX_CIN_OOB 0x80000000u       // added to an offset: the access is out of range, a store is dropped

__device__ __forceinline__ void fx_cin_bstore(__amdgpu_buffer_rsrc_t rs, unsigned voff, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, voff, 0, 0);
}

// Tile arithmetic on MFMA results is plain fmaf: this translation unit is compiled with
// -fno-slp-vectorize (build.py) so that it stays scalar v_fma_f32 — a v_pk_fma_f32 holds the matrix pipe
// ~22 cycles per issue — and the compiler's hazard recogniser sees every read of an MFMA result (it does
// not look inside inline asm).
__device__ __forceinline__ void fx_fmac(float& acc, float x, float y) { acc = fmaf(x, y, acc); }

__device__ __forceinline__ void fx_cin_copy_img(float* lds, const float* img, int n_floats) {
    const float4* src = reinterpret_cast<const float4*>(img);
    float4* dst = reinterpret_cast<float4*>(lds);
#pragma unroll 8
    for (int e = threadIdx.x; e < (n_floats >> 2); e += 64 * FX_CIN_WAVES) dst[e] = src[e];
}

// MQ = quads of m per h (Mi padded to 4*MQ); WF = floats of the LDS image of W (F0 * MQ * 64)
template <int MQ, int WF>
__global__ __launch_bounds__(64 * FX_CIN_WAVES) void k_cin_fwd_mfma(CinArgs a) {
    constexpr int NS = FX_CIN_MFMA_NS, HB = FX_CIN_HB, NT = 64 * FX_CIN_WAVES;
    __shared__ __attribute__((aligned(16))) float Wq[WF];
    const int F0 = a.F0, Mi = a.Mi, O = a.O;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kk = lane >> 4;
    const int nq = F0 * MQ;
    if (a.wimg) {
        fx_cin_copy_img(Wq, a.wimg, nq * 64);
    } else {
        for (int e = threadIdx.x; e < nq * 64; e += NT) Wq[e] = fx_cin_wq_elem(a.W, F0, Mi, O, MQ, e);
    }
    // the K loop runs over whole blocks of 2*HB rows of X0: the image's rows past F0 are zeros
    for (int e = nq * 64 + threadIdx.x; e < fx_cin_rows_padded(F0) * MQ * 64; e += NT) Wq[e] = 0.f;
    float bias4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bias4[i] = (4 * kk + i < O) ? a.bias[4 * kk + i] : 0.f;
    const int64_t stride = (int64_t)gridDim.x * FX_CIN_WAVES * NS;
    const __amdgpu_buffer_rsrc_t rs0 = fx_cin_rsrc(a.X0, a.B, a.x0_ld, F0);
    const __amdgpu_buffer_rsrc_t rsi = fx_cin_rsrc(a.Xi, a.B, a.xi_ld, Mi);
    bool staged = false;
    for (int64_t base = ((int64_t)blockIdx.x * FX_CIN_WAVES + wave) * NS; base < a.B; base += stride) {
        unsigned x0v[NS];
        float xi[NS][MQ];
        float acc[NS][4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s < a.B ? base + s : a.B - 1;      // clamped: stores are guarded
            x0v[s] = (unsigned)((b * a.x0_ld + r) * 4);
            const unsigned xiv = (unsigned)((b * a.xi_ld + kk * 16 + r) * 4);      // row m = 4*mq + kk
#pragma unroll
            for (int mq = 0; mq < MQ; ++mq) xi[s][mq] = fx_cin_bload(rsi, xiv + mq * 256);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[s][i] = 0.f;
        }
        float xa[NS][HB], xb[NS][HB];
        auto loadblk = [&](float (&x)[NS][HB], int h0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned v = x0v[s] + h0 * 64;
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) x[s][hh] = fx_cin_bload(rs0, v + hh * 64);
            }
        };
        auto compute = [&](const float (&x)[NS][HB], int h0) {
            const float* wq = Wq + (h0 * MQ) * 64 + lane;
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                f32x4 T[NS];                         // T_h[o = 4*kk + i][d = r] = sum_m W[o,(h,m)] Xi[m,d]
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) T[s][i] = 0.f;
#pragma unroll
                for (int mq = 0; mq < MQ; ++mq) {
                    const float wa = wq[(hh * MQ + mq) * 64];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
                        T[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, xi[s][mq], T[s], 0, 0, 0);
                }
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int i = 0; i < 4; ++i) fx_fmac(acc[s][i], T[s][i], x[s][hh]);
            }
        };
        loadblk(xa, 0);
        if (!staged) {                   // the operand loads above are in flight behind the W image
            __syncthreads();
            staged = true;
        }
        for (int h0 = 0; h0 < F0; h0 += 2 * HB) {
            loadblk(xb, h0 + HB);
            compute(xa, h0);
            loadblk(xa, h0 + 2 * HB);
            compute(xb, h0 + HB);
        }
        // acc: rows o = 4*kk + i, column d = r
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int64_t b = base + s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int o = 4 * kk + i;
                const float v = acc[s][i] + bias4[i];
                float sum = v;
#pragma unroll
                for (int off = 8; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
                if (b < a.B && o < O) {
                    a.Xn[(b * O + o) * 16 + r] = v;
                    if (a.pool && r == 0) a.pool[b * a.pool_ld + o] = sum;
                }
            }
        }
    }
    if (!staged) __syncthreads();
}

// MT = 16-wide m tiles per h (Mi padded to 16*MT); WF = floats of the LDS image (F0 * MT * 256)
template <int MT, int WF>
__global__ __launch_bounds__(64 * FX_CIN_WAVES) void k_cin_dx_mfma(CinArgs a) {
    constexpr int NS = FX_CIN_MFMA_NS, HB = FX_CIN_HB, NT = 64 * FX_CIN_WAVES;
    __shared__ __attribute__((aligned(16))) float Wb[WF];
    const int F0 = a.F0, Mi = a.Mi, O = a.O;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, kk = lane >> 4;
    const int nt = F0 * MT;
    if (a.wimg) {
        fx_cin_copy_img(Wb, a.wimg + (int64_t)F0 * fx_cin_mq(Mi) * 64, nt * 256);
    } else {
        for (int e = threadIdx.x; e < nt * 256; e += NT) Wb[e] = fx_cin_wb_elem(a.W, F0, Mi, O, MT, e);
    }
    for (int e = nt * 256 + threadIdx.x; e < fx_cin_rows_padded(F0) * MT * 256; e += NT) Wb[e] = 0.f;
    const bool accd = a.acc_dx0 != 0;
    const __amdgpu_buffer_rsrc_t rs0 = fx_cin_rsrc(a.X0, a.B, a.x0_ld, F0);
    const __amdgpu_buffer_rsrc_t rsd = fx_cin_rsrc(a.dX0, a.B, a.dx0_ld, F0);
    const __amdgpu_buffer_rsrc_t rsi = fx_cin_rsrc(a.Xi, a.B, a.xi_ld, Mi);
    const __amdgpu_buffer_rsrc_t rsx = fx_cin_rsrc(a.dXi, a.B, a.dxi_ld, Mi);
    const int64_t stride = (int64_t)gridDim.x * FX_CIN_WAVES * NS;
    bool staged = false;
    for (int64_t base = ((int64_t)blockIdx.x * FX_CIN_WAVES + wave) * NS; base < a.B; base += stride) {
        unsigned x0v[NS], d0v[NS], xiv[NS], div[NS];
        float g[NS][4], xi[NS][MT][4], dxi[NS][MT][4];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const bool live = base + s < a.B;
            const int64_t b = live ? base + s : a.B - 1;
            x0v[s] = (unsigned)((b * a.x0_ld + r) * 4);
            d0v[s] = (unsigned)((b * a.dx0_ld + r) * 4);
            xiv[s] = (unsigned)((b * a.xi_ld + kk * 64 + r) * 4);        // row m = 16*mt + 4*kk + i
            div[s] = (unsigned)((b * a.dxi_ld + kk * 64 + r) * 4) + (live ? 0u : FX_CIN_OOB);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int o = 4 * j + kk;
                float v = 0.f;
                if (o < O) {
                    if (a.dXn) v = a.dXn[(b * O + o) * 16 + r];
                    if (a.dpool) v += a.dpool[b * a.dpool_ld + o];
                }
                g[s][j] = v;
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    xi[s][mt][i] = fx_cin_bload(rsi, xiv[s] + mt * 1024 + i * 64);
                    dxi[s][mt][i] = 0.f;
                }
        }
        // X0 rows travel in register blocks of HB, two in flight; when dX0 accumulates, the values already
        // there are fetched at the head of the block that finishes them
        float xa[NS][HB], xb[NS][HB];
        auto loadblk = [&](float (&x)[NS][HB], int h0) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned vx = x0v[s] + h0 * 64;
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) x[s][hh] = fx_cin_bload(rs0, vx + hh * 64);
            }
        };
        auto compute = [&](const float (&x)[NS][HB], int h0) {
            float old[NS][HB];
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned vd = d0v[s] + h0 * 64;
#pragma unroll
                for (int hh = 0; hh < HB; ++hh) old[s][hh] = accd ? fx_cin_bload(rsd, vd + hh * 64) : 0.f;
            }
#pragma unroll
            for (int hh = 0; hh < HB; ++hh) {
                float dx0[NS];
#pragma unroll
                for (int s = 0; s < NS; ++s) dx0[s] = 0.f;
#pragma unroll
                for (int mt = 0; mt < MT; ++mt) {
                    const float* wb = Wb + ((h0 + hh) * MT + mt) * 256 + lane;
                    f32x4 T[NS];
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int i = 0; i < 4; ++i) T[s][i] = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float wa = wb[j * 64];
#pragma unroll
                        for (int s = 0; s < NS; ++s)
                            T[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, g[s][j], T[s], 0, 0, 0);
                    }
                    // T[s][i] = T[(h, m = 16*mt + 4*kk + i), d = r]  (zeros for the image's rows past F0)
#pragma unroll
                    for (int s = 0; s < NS; ++s)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            fx_fmac(dx0[s], T[s][i], xi[s][mt][i]);
                            fx_fmac(dxi[s][mt][i], T[s][i], x[s][hh]);
                        }
                    if (MT > 1) __builtin_amdgcn_sched_barrier(0);     // bounds the live set (no spills)
                }
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    float v = dx0[s];
                    v += __shfl_xor(v, 16, 64);
                    v += __shfl_xor(v, 32, 64);
                    old[s][hh] += v;
                }
                __builtin_amdgcn_sched_barrier(0);       // one row's live set at a time (no spills)
            }
            // the block's dX0 rows, after its arithmetic: lanes kk == 0 of live samples, rows < F0 (the
            // resource ends at row F0 of the last sample; other samples' rows past F0 are masked here)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const unsigned vd = d0v[s] + h0 * 64 + ((kk == 0 && base + s < a.B) ? 0u : FX_CIN_OOB);
#pragma unroll
                for (int hh = 0; hh < HB; ++hh)
                    fx_cin_bstore(rsd, h0 + hh < F0 ? vd + hh * 64 : FX_CIN_OOB, old[s][hh]);
            }
        };
        loadblk(xa, 0);
        if (!staged) {
            __syncthreads();
            staged = true;
        }
        for (int h0 = 0; h0 < F0; h0 += 2 * HB) {
            loadblk(xb, h0 + HB);
            compute(xa, h0);
            loadblk(xa, h0 + 2 * HB);
            compute(xb, h0 + HB);
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = 16 * mt + 4 * kk + i;
                    fx_cin_bstore(rsx, m < Mi ? div[s] + mt * 1024 + i * 64 : FX_CIN_OOB, dxi[s][mt][i]);
                }
    }
    if (!staged) __syncthreads();
}

#define FX_CIN_DW_SB 4     // samples staged per barrier in the dW kernel

// HPW = rows h of X0 per wave = ceil(F0 / 8): wave w owns h = w, w + 8, ... and their MT column tiles
template <int MT, int HPW>
__global__ __launch_bounds__(64 * FX_CIN_WAVES) void k_cin_dw_mfma(CinArgs a) {
    constexpr int SB = FX_CIN_DW_SB, NT = 64 * FX_CIN_WAVES, NW = FX_CIN_WAVES;
    constexpr int X0Q = (48 * 16 + NT - 1) / NT, XIQ = (MT * 256 + NT - 1) / NT;
    __shared__ float x0s[2][SB][48 * 16];  // a sample's X0 [F0 <= 40][16], zero rows up to 48; double-buffered
    __shared__ float gs[2][SB][16 * 17];   // g[o][d], rows padded to 17 floats like Xi's
    __shared__ float xis[2][SB][48 * 17];  // Xi rows padded to 17 floats (conflict-free column reads)
    const int F0 = a.F0, Mi = a.Mi, O = a.O, C = F0 * Mi;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kk = lane >> 4;
    f32x4 acc[HPW][MT];
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[hh][mt][i] = 0.f;
    float db = 0.f;                        // wave 0: sum over (b, d = 4*j + kk) of g[o = r][d]
    // SB samples' operands travel global -> registers (issued before the MFMA section) -> LDS (after
    // it); sample sb of round k is b = (k * SB + sb) * gridDim.x + blockIdx.x, absent ones are zeros
    float px0[SB][X0Q], pxi[SB][XIQ], pg[SB];
    auto fetch = [&](int64_t k) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
            const int64_t b = (k * SB + sb) * gridDim.x + blockIdx.x;
            const bool live = b < a.B;
#pragma unroll
            for (int q = 0; q < X0Q; ++q) {
                const int e = tid + NT * q;
                px0[sb][q] = (live && e < F0 * 16) ? a.X0[b * a.x0_ld + e] : 0.f;
            }
#pragma unroll
            for (int q = 0; q < XIQ; ++q) {
                const int m = (tid + NT * q) >> 4;
                pxi[sb][q] = (live && m < Mi) ? a.Xi[b * a.xi_ld + m * 16 + (tid & 15)] : 0.f;
            }
            const int o = tid >> 4;
            float v = 0.f;
            if (live && o < O) {           // O <= 16: threads 0..255
                if (a.dXn) v = a.dXn[b * O * 16 + tid];
                if (a.dpool) v += a.dpool[b * a.dpool_ld + o];
            }
            pg[sb] = v;
        }
    };
    auto put = [&](int buf) {
#pragma unroll
        for (int sb = 0; sb < SB; ++sb) {
#pragma unroll
            for (int q = 0; q < X0Q; ++q)
                if (tid + NT * q < 48 * 16) x0s[buf][sb][tid + NT * q] = px0[sb][q];
#pragma unroll
            for (int q = 0; q < XIQ; ++q) {
                const int m = (tid + NT * q) >> 4;
                if (m < MT * 16) xis[buf][sb][m * 17 + (tid & 15)] = pxi[sb][q];
            }
            if (tid < 256) gs[buf][sb][(tid >> 4) * 17 + (tid & 15)] = pg[sb];
        }
    };
    const int64_t per_round = (int64_t)SB * gridDim.x;
    const int64_t rounds = (a.B - blockIdx.x + per_round - 1) / per_round;   // blockIdx.x >= B: 0 rounds
    int buf = 0;
    if (rounds > 0) {
        fetch(0);
        put(0);
    }
    __syncthreads();
    for (int64_t k = 0; k < rounds; ++k) {
        if (k + 1 < rounds) fetch(k + 1);
#pragma unroll 1
        for (int sb = 0; sb < SB; ++sb) {
            float ga[4], xr[MT][4], ah[HPW][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ga[j] = gs[buf][sb][r * 17 + 4 * j + kk];
                if (wave == 0) db += ga[j];
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int j = 0; j < 4; ++j) xr[mt][j] = xis[buf][sb][(16 * mt + r) * 17 + 4 * j + kk];
            // A fragments of the whole sweep first: (g X0_h)[o = r][d = 4*j + kk]; rows h >= F0 are zeros
            const float* xw = x0s[buf][sb] + wave * 16 + kk;
#pragma unroll
            for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
                for (int j = 0; j < 4; ++j) ah[hh][j] = ga[j] * xw[hh * NW * 16 + 4 * j];
            // consecutive MFMAs hit different accumulators
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int hh = 0; hh < HPW; ++hh)
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt)
                        acc[hh][mt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ah[hh][j], xr[mt][j], acc[hh][mt],
                                                                          0, 0, 0);
        }
        if (k + 1 < rounds) put(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    // partial[blockIdx][o * C + h * Mi + m]: D rows o = 4*kk + i, column m = 16*mt + r
    float* part = a.partial + (int64_t)blockIdx.x * a.partial_ld;
#pragma unroll
    for (int hh = 0; hh < HPW; ++hh) {
        const int h = wave + NW * hh;
        if (h < F0) {
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int m = mt * 16 + r;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = 4 * kk + i;
                    if (o < O && m < Mi) part[(int64_t)o * C + h * Mi + m] = acc[hh][mt][i];
                }
            }
        }
    }
    if (wave == 0) {
        db += __shfl_xor(db, 16, 64);
        db += __shfl_xor(db, 32, 64);
        if (kk == 0 && r < O) part[(int64_t)O * C + r] = db;
    }
}

// the MFMA class: D = 16, O <= 16, F0 <= 40, Mi <= 40

static bool fx_cin_offsets_fit(int64_t B, int64_t ld) {      // byte offsets of the buffer loads: 31 bits
    return B * ld * 4 < ((int64_t)1 << 31);
}

bool fx_cin_mfma_shape(int32_t F0, int32_t Mi, int32_t D, int32_t O) {
    return fx_cin_mfma_mode() && D == 16 && O >= 1 && O <= 16 && F0 >= 1 && F0 <= 40 && Mi >= 1 && Mi <= 40;
}

int64_t fx_cin_mfma_wimg_floats(int32_t F0, int32_t Mi) {
    return (int64_t)F0 * fx_cin_mq(Mi) * 64 + (int64_t)F0 * fx_cin_mt(Mi) * 256;
}

void fx_cin_mfma_pack_w(const CinPackArgs& pa, hipStream_t s) {
    int64_t n = 0;
    for (int i = 0; i < pa.n; ++i) {
        const int64_t ni = fx_cin_mfma_wimg_floats(pa.F0[i], pa.Mi[i]);
        n = ni > n ? ni : n;
    }
    hipLaunchKernelGGL(k_cin_pack_w, dim3((unsigned)fx_ceil_div(n, 256), (unsigned)pa.n), dim3(256), 0, s, pa);
}

static unsigned fx_cin_sample_grid(int64_t B) {
    const int64_t per_wg = FX_CIN_WAVES * FX_CIN_MFMA_NS;
    const int64_t g = fx_ceil_div(B, per_wg);
    return (unsigned)(g < 256 ? g : 256);
}

bool fx_cin_mfma_fwd(const CinArgs& a, hipStream_t s) {
    if (!fx_cin_offsets_fit(a.B, a.x0_ld) || !fx_cin_offsets_fit(a.B, a.xi_ld)) return false;
    const dim3 grid(fx_cin_sample_grid(a.B)), block(64 * FX_CIN_WAVES);
    if (a.Mi <= 16)
        hipLaunchKernelGGL((k_cin_fwd_mfma<4, 40 * 4 * 64>), grid, block, 0, s, a);
    else
        hipLaunchKernelGGL((k_cin_fwd_mfma<10, 40 * 10 * 64>), grid, block, 0, s, a);
    return true;
}

bool fx_cin_mfma_bwd(const CinArgs& a, hipStream_t s) {
    if (!fx_cin_offsets_fit(a.B, a.x0_ld) || !fx_cin_offsets_fit(a.B, a.dx0_ld) ||
        !fx_cin_offsets_fit(a.B, a.xi_ld) || !fx_cin_offsets_fit(a.B, a.dxi_ld))
        return false;
    const dim3 grid(fx_cin_sample_grid(a.B)), block(64 * FX_CIN_WAVES);
    if (a.Mi <= 16) {
        hipLaunchKernelGGL((k_cin_dx_mfma<1, 40 * 1 * 256>), grid, block, 0, s, a);
        hipLaunchKernelGGL((k_cin_dw_mfma<1, 5>), dim3(256), block, 0, s, a);
    } else {
        hipLaunchKernelGGL((k_cin_dx_mfma<3, 40 * 3 * 256>), grid, block, 0, s, a);
        hipLaunchKernelGGL((k_cin_dw_mfma<3, 5>), dim3(256), block, 0, s, a);
    }
    return true;
}
