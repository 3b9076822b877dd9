// Micro-benchmark: how fast can one CU stream v_mfma_f32_32x32x2_f32 under the access patterns of
// k_gemm_f32?  hipcc --offload-arch=gfx950 -O3 -o mfma_stream mfma_stream.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 129
// MODE 0: MFMA only.  1: + fragment reads from LDS (k-major, ds_read_b32), software pipelined
// 2: + one barrier per 16 groups.  3: + 32 ds_write_b32 per thread per tile (transposing refill)
template <int MODE, int WAVES_PER_EU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES_PER_EU, WAVES_PER_EU)))
void k(float* out, int iters) {
    __shared__ float As[2][32 * LD], Bs[2][32 * LD];
    for (int i = threadIdx.x; i < 2 * 32 * LD; i += 256) { (&As[0][0])[i] = 1.f; (&Bs[0][0])[i] = 0.5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float fa[2][2] = {{1.f, 1.f}, {1.f, 1.f}}, fb[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
    float4 st[8];
    for (int p = 0; p < 8; ++p) st[p] = make_float4(1.f, 2.f, 3.f, 4.f);
    for (int t = 0; t < iters; ++t) {
        const int cur = t & 1;
        const float* as = As[cur] + half * LD + wm * 64 + l31;
        const float* bs = Bs[cur] + half * LD + wn * 64 + l31;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int c = g & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i], fb[c][j], acc[i][j], 0, 0, 0);
            if (MODE >= 1) {
                const int kk = (2 * g + 4) & 31;
#pragma unroll
                for (int i = 0; i < 2; ++i) fa[c][i] = as[kk * LD + 32 * i];
#pragma unroll
                for (int j = 0; j < 2; ++j) fb[c][j] = bs[kk * LD + 32 * j];
            }
            if (MODE >= 3 && g >= 2 && g < 10) {
                const int p = g - 2;
                const int q = threadIdx.x + 256 * (p & 3);
                float* T = (p < 4 ? As[cur ^ 1] : Bs[cur ^ 1]);
                const int r = q >> 3, kq = (q & 7) << 2;
                T[(kq + 0) * LD + r] = st[p].x;
                T[(kq + 1) * LD + r] = st[p].y;
                T[(kq + 2) * LD + r] = st[p].z;
                T[(kq + 3) * LD + r] = st[p].w;
            }
            if (MODE >= 2 && g == 11) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + fa[0][0] + fb[1][1];
}

template <int MODE, int W>
void run(const char* name, float* out, int wg_per_cu) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, W>), dim3(grid), dim3(256), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, W>), dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-34s mode %d waves/eu %d wg/cu %d : %8.3f ms  %7.1f TFLOP/s\n", name, MODE, W, wg_per_cu, ms, fl / ms / 1e9);
}

int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    run<0, 1>("mfma only (agpr form)", out, 1);
    run<0, 2>("mfma only (vgpr form)", out, 1);
    run<0, 2>("mfma only (vgpr form)", out, 2);
    run<1, 1>("+frag reads", out, 1);
    run<1, 2>("+frag reads", out, 1);
    run<1, 2>("+frag reads", out, 2);
    run<2, 1>("+barrier", out, 1);
    run<2, 2>("+barrier", out, 2);
    run<3, 1>("+lds refill writes", out, 1);
    run<3, 2>("+lds refill writes", out, 1);
    run<3, 2>("+lds refill writes", out, 2);
    return 0;
}
