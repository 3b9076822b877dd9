// Single-wave issue rate of v_mfma_f32_16x16x4_f32 under the operand patterns of the CIN kernels.
// build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define N_ITER 2000

template <int MODE, int NACC>
__global__ __launch_bounds__(512) void k(const float* in, float* out) {
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = in[i & 255];
    __syncthreads();
    f32x4 acc[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s) acc[s] = f32x4{0, 0, 0, 0};
    float a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
    float x[NACC];
#pragma unroll
    for (int s = 0; s < NACC; ++s) x[s] = in[s * 64 + (threadIdx.x & 63)];
    if (MODE & 8) {                       // products of group u+1 computed while group u's MFMAs issue
        float p[2][NACC];
#pragma unroll
        for (int s = 0; s < NACC; ++s) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[0][s]) : "v"(x[s]), "v"(b));
        for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int s = 0; s < NACC; ++s) {
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p[(u + 1) & 1][s]) : "v"(x[s]), "v"(p[u & 1][s]));
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, p[u & 1][s], acc[s], 0, 0, 0);
                }
            }
        }
    } else if (MODE & 16) {               // an unrelated VALU op between MFMAs
        float y[NACC];
#pragma unroll
        for (int s = 0; s < NACC; ++s) y[s] = x[s];
        for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
#pragma unroll
                for (int s = 0; s < NACC; ++s) {
                    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(y[s]) : "v"(y[s]), "v"(b));
                    acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[s], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < NACC; ++s) acc[s][0] += y[s];
    } else
    for (int it = 0; it < N_ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            float wa = a;
            if (MODE & 2) wa = lds[((it * 8 + u) & 63) * 64 + (threadIdx.x & 63)];
#pragma unroll
            for (int s = 0; s < NACC; ++s) {
                float bb = b;
                if (MODE & 1) bb = x[s] * b;            // v_mul feeding the MFMA
                if (MODE & 4) asm volatile("v_mul_f32 %0, %1, %2" : "=v"(bb) : "v"(x[s]), "v"(wa));
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa, bb, acc[s], 0, 0, 0);
            }
            if (MODE & 1) b += 1.0f;
        }
    }
    float r = 0;
#pragma unroll
    for (int s = 0; s < NACC; ++s) r += acc[s][0] + acc[s][1] + acc[s][2] + acc[s][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int NACC>
void run(const char* name, int threads, const float* in, float* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(threads), 0, 0, in, out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(threads), 0, 0, in, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double per_wave = (double)N_ITER * 8 * NACC;                 // MFMAs per wave
    double waves_per_simd = threads / 256.0;
    double ns = ms / 5 * 1e6 / (per_wave * waves_per_simd);     // ns per MFMA per SIMD
    printf("%-44s threads %4d  %.2f ns/MFMA/SIMD = %.1f cyc @2.4GHz  (%.1f TF)\n", name, threads, ns, ns * 2.4,
           2048.0 / ns * 1024 / 1e3);
}

int main() {
    float *in, *out;
    hipMalloc(&in, 4096 * 4); hipMalloc(&out, 256 * 1024 * 4);
    float h[4096]; for (int i = 0; i < 4096; ++i) h[i] = 1e-3f * (i % 17);
    hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 4>("mfma only, 4 acc", 256, in, out);
        run<0, 4>("mfma only, 4 acc", 512, in, out);
        run<0, 8>("mfma only, 8 acc", 256, in, out);
        run<1, 4>("v_mul -> mfma, 4 acc", 256, in, out);
        run<1, 4>("v_mul -> mfma, 4 acc", 512, in, out);
        run<4, 4>("asm v_mul -> mfma, 4 acc", 256, in, out);
        run<2, 4>("ds_read A per 4 mfma", 256, in, out);
        run<3, 4>("ds_read A + v_mul", 256, in, out);
        run<3, 4>("ds_read A + v_mul", 512, in, out);
        run<3, 2>("ds_read A + v_mul, 2 acc", 256, in, out);
        run<8, 4>("v_mul one group ahead, 4 acc", 256, in, out);
        run<8, 4>("v_mul one group ahead, 4 acc", 512, in, out);
        run<8, 8>("v_mul one group ahead, 8 acc", 256, in, out);
        run<16, 4>("unrelated v_mul between mfma, 4 acc", 256, in, out);
        run<16, 4>("unrelated v_mul between mfma, 4 acc", 512, in, out);
    }
    return 0;
}
