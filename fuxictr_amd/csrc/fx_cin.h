// fx_cin.h — shared between fx_cin.hip (fp32 VALU kernels, C-ABI entry points) and fx_cin_mfma.hip
// (the matrix-core kernels of the D = 16 shape class).
#pragma once
#include "fx_common.h"

struct CinArgs {
    const float* X0; int64_t x0_ld;
    const float* Xi; int64_t xi_ld;
    const float* W;          // [O, C]   C = F0 * Mi
    const float* bias;       // [O]
    float* Xn;               // [B, O, D]
    float* pool; int64_t pool_ld;   // pool[b*pool_ld + o] = sum_d Xn[b,o,d]
    const float* dXn;        // [B, O, D] or null
    const float* dpool; int64_t dpool_ld;
    float* dX0; int64_t dx0_ld;
    float* dXi; int64_t dxi_ld;
    float* partial;          // [G][O*C + O], row stride partial_ld
    int64_t partial_ld;
    int64_t B;
    int32_t F0, Mi, D, O, acc_dx0;
    const float* wimg;       // fx_cin_pack_w's LDS images of W (MFMA kernels), or null
};

// fx_cin_mfma.hip
bool fx_cin_mfma_shape(int32_t F0, int32_t Mi, int32_t D, int32_t O);   // the shape class (and FX_CIN_MFMA != 0)
int64_t fx_cin_mfma_wimg_floats(int32_t F0, int32_t Mi);
struct CinPackArgs {        // up to FX_CIN_PACK_MAX layers' images in one launch
    const float* W[4];
    float* img[4];
    int32_t F0[4], Mi[4], O[4];
    int32_t n;
};
#define FX_CIN_PACK_MAX 4
void fx_cin_mfma_pack_w(const CinPackArgs& pa, hipStream_t s);
// false: not launched (byte offsets beyond the buffer loads' 31 bits) -> the VALU kernels take the call
bool fx_cin_mfma_fwd(const CinArgs& a, hipStream_t s);
bool fx_cin_mfma_bwd(const CinArgs& a, hipStream_t s);
