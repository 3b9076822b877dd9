// fx_series.hip — builder of the Adam series table (round 6).  Mathematics and layout: include/fxctr.h
// (fx_adam_series_build); consumer: fx_series_move in fx_common.h.
//
// What it replaces: the step-by-step replay of a row's missed zero-gradient Adam steps
// (torch.optim.Adam over a dense-gradient nn.Embedding moves every row every step: rank_model.py:322,
// torch_utils.py:72-76).  For a row last updated at step t the move of step t + i factors into a row part
// (lr m / sqrt(v), c = eps / sqrt(v)) and the pair
//     w_i(t) = b1^i / (1 - B1^(t+i)),      g_i(t) = b2^(i/2) / sqrt(1 - B2^(t+i))
// (b: the fp32 images the tensor ops decay the moments with, B: the python-side doubles of the bias corrections)
// that every row of that t shares.  One table entry per t holds  F(t; c) = sum_{i>=1} w_i / (g_i + c)  as
// nseg segments of i (1 | 24 | 8,24,64 | 2,4,...,128 are the cut ladders), each expanded about its weighted
// mean g_s = sum w g / sum w:
//     1 / (g_i + c) = (1/g_s) sum_n (-d_i)^n y^(n+1),   d_i = g_i / g_s - 1,  y = g_s / (g_s + c) in (0, 1]
//     sum over the segment = (y/g_s) (mu0 + y^2 mu2 - y^3 mu3 + ... + y^6 mu6),   mu_n = sum_i w_i d_i^n
// (mu1 = 0 by the choice of g_s).  |d_i| <= the segment's spread, y <= 1: the truncation does not depend on c,
// so the entry is checked HERE, once, against the directly summed series at eight values of c from 0 to
// 100 g_1, and the worst relative error of the whole table is left in the header for the host to read.
// With torch's default betas: one segment from t = 128 on (error <= 3e-8), up to eight below (<= 1.2e-7 at
// t = 1, where g_i falls from 22 to 3 over the first hundred steps).
#include "fx_common.h"

#define FX_SER_IMAX 512          // terms of the "infinite" sum: (b1/sqrt(b2))^512 ~ 5e-24 for the defaults
#define FX_SER_TOL 3.0e-8        // an early entry takes the shortest ladder that reaches this

struct SerBetas {
    double b1, b2, sb2;      // fp32 images: what the tensor ops multiply the moments by (the decay b1^i, b2^(i/2))
    double c1, c2;           // the python-side doubles: the bias corrections 1 - B^(t+i) (fx_beta_f64)
};

// w_i, g_i for i = i0 .. i1 of entry t, handed to `f(i, w, g)`; powers by recurrence in fp64
template <class F>
__device__ __forceinline__ void fx_ser_terms(const SerBetas& be, int t, int i0, int i1, F f) {
    double bi = pow(be.b1, (double)(i0 - 1)), hi = pow(be.sb2, (double)(i0 - 1));
    double q1 = pow(be.c1, (double)(t + i0 - 1)), q2 = pow(be.c2, (double)(t + i0 - 1));
    for (int i = i0; i <= i1; ++i) {
        bi *= be.b1; hi *= be.sb2; q1 *= be.c1; q2 *= be.c2;
        const double w = bi / (1.0 - q1);
        const double g = hi / sqrt(1.0 - q2);
        f(i, w, g);
    }
}

__device__ __forceinline__ void fx_ser_cuts(int nseg, int (&hi)[8]) {
    // last term of every segment
    if (nseg == 1) { hi[0] = FX_SER_IMAX; }
    else if (nseg == 2) { hi[0] = 24; hi[1] = FX_SER_IMAX; }
    else if (nseg == 4) { hi[0] = 8; hi[1] = 24; hi[2] = 64; hi[3] = FX_SER_IMAX; }
    else { hi[0] = 2; hi[1] = 4; hi[2] = 8; hi[3] = 16; hi[4] = 32; hi[5] = 64; hi[6] = 128; hi[7] = FX_SER_IMAX; }
}

// builds the nseg segments of entry t into seg[s][0..6] = {g_s, c0, c2, c3, c4, c5, c6}; returns the worst
// relative error of the entry against the direct sum over the probe values of c
__device__ double fx_ser_build_entry(const SerBetas& be, int t, int nseg, double (&seg)[8][7]) {
    int hi[8];
    fx_ser_cuts(nseg, hi);
    int lo = 1;
    double g1 = 0.0;
    for (int s = 0; s < nseg; ++s) {
        double W = 0.0, WG = 0.0;
        fx_ser_terms(be, t, lo, hi[s], [&](int i, double w, double g) {
            W += w; WG += w * g;
            if (i == 1) g1 = g;
        });
        const double gs = WG / W;
        double mu[7] = {0, 0, 0, 0, 0, 0, 0};
        fx_ser_terms(be, t, lo, hi[s], [&](int, double w, double g) {
            const double d = g / gs - 1.0;
            double pw = w;
#pragma unroll
            for (int n = 0; n < 7; ++n) { mu[n] += pw; pw *= d; }
        });
        seg[s][0] = gs;
        seg[s][1] = mu[0] / gs;
        seg[s][2] = mu[2] / gs;
        seg[s][3] = -mu[3] / gs;
        seg[s][4] = mu[4] / gs;
        seg[s][5] = -mu[5] / gs;
        seg[s][6] = mu[6] / gs;
        lo = hi[s] + 1;
    }
    // probe: c = r g_1, r in {0, .03, .1, .3, 1, 3, 10, 100}
    const double rr[8] = {0.0, 0.03, 0.1, 0.3, 1.0, 3.0, 10.0, 100.0};
    double ex[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    double wlast = 0.0, wtot = 0.0;
    fx_ser_terms(be, t, 1, FX_SER_IMAX, [&](int i, double w, double g) {
#pragma unroll
        for (int q = 0; q < 8; ++q) ex[q] += w / (g + rr[q] * g1);
        wtot += w;
        if (i == FX_SER_IMAX) wlast = w;
    });
    double worst = 0.0;
    for (int q = 0; q < 8; ++q) {
        const double c = rr[q] * g1;
        double ap = 0.0;
        for (int s = 0; s < nseg; ++s) {
            const double y = seg[s][0] / (seg[s][0] + c);
            double p = seg[s][6];
            p = seg[s][5] + y * p;
            p = seg[s][4] + y * p;
            p = seg[s][3] + y * p;
            p = seg[s][2] + y * p;
            p = seg[s][1] + y * y * p;
            ap += y * p;
        }
        const double e = fabs(ap - ex[q]) / ex[q];
        worst = e > worst ? e : worst;
    }
    // the sum has not converged inside FX_SER_IMAX terms (b1 / sqrt(b2) too close to 1): unusable
    if (!(wlast <= 1e-12 * wtot)) worst = 1.0e30;
    return worst;
}

__global__ __launch_bounds__(64) void k_series_build(fx_scalars* scal, int tcap) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= tcap) return;
    SerBetas be;
    be.b1 = (double)scal->beta1;
    be.b2 = (double)scal->beta2;
    be.sb2 = sqrt(be.b2);
    be.c1 = fx_beta_f64(scal->beta1);
    be.c2 = fx_beta_f64(scal->beta2);
    float* base = reinterpret_cast<float*>(scal) + 16;
    float* tab = base + FX_SER_HDR;
    double seg[8][7];
    double err = 0.0;
    int nseg = 1;
    if (t < FX_SERIES_EARLY) {
        for (nseg = 1; nseg <= 8; nseg *= 2) {
            err = fx_ser_build_entry(be, t, nseg, seg);
            if (err <= FX_SER_TOL || nseg == 8) break;
        }
        float* e = tab + (int64_t)t * FX_SER_ENTRYW;
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int j = 0; j < 7; ++j) e[s * FX_SER_SEGW + j] = s < nseg ? (float)seg[s][j] : 0.f;
            e[s * FX_SER_SEGW + 7] = 0.f;
        }
        reinterpret_cast<int32_t*>(e)[7] = nseg;
    } else {
        err = fx_ser_build_entry(be, t, 1, seg);
        float* e = tab + (int64_t)FX_SERIES_EARLY * FX_SER_ENTRYW + (int64_t)(t - FX_SERIES_EARLY) * FX_SER_SEGW;
#pragma unroll
        for (int j = 0; j < 7; ++j) e[j] = (float)seg[0][j];
        e[7] = 0.f;
    }
    // entry 0 is never read for a live row (a row with last_step = 0 has no moments); its error does not count
    if (t > 0) {
        const float ef = err < 1.0e30 ? (float)err : 1.0e30f;
        atomicMax(reinterpret_cast<int32_t*>(base) + 2, __float_as_int(ef));     // (non-negative floats order as ints)
    }
}

__global__ void k_series_publish(fx_scalars* scal, int tcap) {
    int32_t* hdr = reinterpret_cast<int32_t*>(scal) + 16;
    hdr[0] = 0x46585352;          // "FXSR"
    hdr[1] = tcap;
    scal->series_tcap = tcap;
}

extern "C" int64_t fx_adam_series_words(int32_t tcap) {
    if (tcap <= FX_SERIES_EARLY) return 0;
    return (int64_t)FX_SER_HDR + (int64_t)FX_SERIES_EARLY * FX_SER_ENTRYW +
           (int64_t)(tcap - FX_SERIES_EARLY) * FX_SER_SEGW;
}

extern "C" int fx_adam_series_build(fx_scalars* scal, int32_t tcap, fx_stream_t stream) {
    FX_CHECK_ARG(scal, "fx_adam_series_build: null scal");
    FX_CHECK_ARG(tcap > FX_SERIES_EARLY && tcap <= (1 << 22),
                 "fx_adam_series_build: tcap=%d outside (%d, %d]", tcap, FX_SERIES_EARLY, 1 << 22);
    FX_CHECK_ARG((reinterpret_cast<uintptr_t>(scal) & 31) == 0,
                 "fx_adam_series_build: scal must be 32-byte aligned (the entries are read as float4)");
    hipStream_t s = fx_hip_stream(stream);
    FX_CHECK_HIP(hipMemsetAsync(reinterpret_cast<char*>(scal) + 64, 0, FX_SER_HDR * 4, s));
    hipLaunchKernelGGL(k_series_build, dim3((unsigned)fx_ceil_div(tcap, 64)), dim3(64), 0, s, scal, (int)tcap);
    FX_CHECK_LAUNCH();
    hipLaunchKernelGGL(k_series_publish, dim3(1), dim3(1), 0, s, scal, (int)tcap);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
