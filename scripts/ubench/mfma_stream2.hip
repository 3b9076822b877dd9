// Micro-benchmark 2: same LDS work as k_gemm_f32's tile loop, differently scheduled.
// MODE 4: k-major b32 layout (4 frag reads per group, 32 ds_write_b32 per tile), ONE LDS op per MFMA
// MODE 5: [r][k] layout (row stride 36): 16 ds_read_b128 + 8 ds_write_b128 per tile, spread out
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define LD 129
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ float sel(const float4& v, int e) { return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w; }

// MODE 6/7: mode 4 + 8 global_load_dwordx4 per tile (tile t+2 into a second register set), one per
//   MFMA in groups 0..1 (6) or all 8 in a clump before group 0 (7), stores select-zeroed
template <int MODE, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
void k3(float* out, const float* __restrict__ src, int ld, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][32 * LD], Bs[2][32 * LD];
    for (int i = threadIdx.x; i < 2 * 32 * LD; i += 256) { (&As[0][0])[i] = 1.f; (&Bs[0][0])[i] = 0.5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 st[2][8];
    for (int c = 0; c < 2; ++c) for (int p = 0; p < 8; ++p) st[c][p] = make_float4(1.f, 2.f, 3.f, 4.f);
    int rc[8];
    for (int p = 0; p < 8; ++p) rc[p] = ((blockIdx.x * 128 + (threadIdx.x >> 3) + 32 * (p & 3)) % 4096) * ld + ((threadIdx.x & 7) << 2);
    const int kend = ld;
    float fa[2][2] = {{1.f, 1.f}, {1.f, 1.f}}, fb[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
    auto body = [&](int t, auto par) {
        constexpr int P = decltype(par)::value;
        const int cur = t & 1;
        const float* as = As[cur] + half * LD + wm * 64 + l31;
        const float* bs = Bs[cur] + half * LD + wn * 64 + l31;
        const int k0 = ((t + 2) * 32) % (ld - 32);
        if (MODE == 7) {
#pragma unroll
            for (int p = 0; p < 8; ++p) {
                const int k = k0 + ((threadIdx.x & 7) << 2);
                const int kc = k < kend ? k : kend - 4;
                st[P][p] = *reinterpret_cast<const float4*>(src + rc[p] + kc);
            }
        }
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int c = g & 1, kk = (2 * g + 4) & 31, p = (g - 4) & 7;
            const int q = threadIdx.x + 256 * (p & 3);
            float* T = (p < 4 ? As[cur ^ 1] : Bs[cur ^ 1]);
            const int r = q >> 3, kq = (q & 7) << 2;
            const bool wr = (g >= 4 && g < 12);
            const bool ok = (rc[p] & 1) == 0;
            auto ldg = [&](int pl) {
                if (MODE == 6 && g < 2) {
                    const int k = k0 + ((threadIdx.x & 7) << 2);
                    const int kc = k < kend ? k : kend - 4;
                    st[P][pl] = *reinterpret_cast<const float4*>(src + rc[pl] + kc);
                }
            };
            MF(fa[c][0], fb[c][0], acc[0][0]);
            const float na0 = as[kk * LD];
            if (wr) T[(kq + 0) * LD + r] = ok ? st[P ^ 1][p].x : 0.f;
            ldg(g * 4 + 0);
            SB();
            MF(fa[c][0], fb[c][1], acc[0][1]);
            const float na1 = as[kk * LD + 32];
            if (wr) T[(kq + 1) * LD + r] = ok ? st[P ^ 1][p].y : 0.f;
            ldg(g * 4 + 1);
            SB();
            MF(fa[c][1], fb[c][0], acc[1][0]);
            const float nb0 = bs[kk * LD];
            if (wr) T[(kq + 2) * LD + r] = ok ? st[P ^ 1][p].z : 0.f;
            ldg(g * 4 + 2);
            SB();
            MF(fa[c][1], fb[c][1], acc[1][1]);
            const float nb1 = bs[kk * LD + 32];
            if (wr) T[(kq + 3) * LD + r] = ok ? st[P ^ 1][p].w : 0.f;
            ldg(g * 4 + 3);
            if (g == 13) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            SB();
            fa[c][0] = na0; fa[c][1] = na1; fb[c][0] = nb0; fb[c][1] = nb1;
        }
    };
    for (int t = 0; t < iters; t += 2) {
        body(t, std::integral_constant<int, 0>{});
        body(t + 1, std::integral_constant<int, 1>{});
    }
    float s = fa[0][0] + fb[1][1];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int W>
void run3(const char* name, float* out, const float* src, int wg_per_cu) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k3<MODE, W>), dim3(grid), dim3(256), 0, 0, out, src, 4096, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k3<MODE, W>), dim3(grid), dim3(256), 0, 0, out, src, 4096, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-34s mode %d waves/eu %d wg/cu %d : %8.3f ms  %7.1f TFLOP/s\n", name, MODE, W, wg_per_cu, ms, fl / ms / 1e9);
}

template <int MODE, int W>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(W, W)))
void k2(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float As[2][128 * 36], Bs[2][128 * 36];
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) { (&As[0][0])[i] = 1.f; (&Bs[0][0])[i] = 0.5f; }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31, wm = wave >> 1, wn = wave & 1;
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 st[8];
    for (int p = 0; p < 8; ++p) st[p] = make_float4(1.f, 2.f, 3.f, 4.f);
    float res = 0.f;
    if (MODE == 4) {
        float fa[2][2] = {{1.f, 1.f}, {1.f, 1.f}}, fb[2][2] = {{1.f, 1.f}, {1.f, 1.f}};
        for (int t = 0; t < iters; ++t) {
            const int cur = t & 1;
            const float* as = As[cur] + half * LD + wm * 64 + l31;
            const float* bs = Bs[cur] + half * LD + wn * 64 + l31;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int c = g & 1, kk = (2 * g + 4) & 31, p = (g - 2) & 7;
                const int q = threadIdx.x + 256 * (p & 3);
                float* T = (p < 4 ? As[cur ^ 1] : Bs[cur ^ 1]);
                const int r = q >> 3, kq = (q & 7) << 2;
                const bool wr = (g >= 2 && g < 10);
                MF(fa[c][0], fb[c][0], acc[0][0]);
                const float na0 = as[kk * LD];
                if (wr) T[(kq + 0) * LD + r] = st[p].x;
                SB();
                MF(fa[c][0], fb[c][1], acc[0][1]);
                const float na1 = as[kk * LD + 32];
                if (wr) T[(kq + 1) * LD + r] = st[p].y;
                SB();
                MF(fa[c][1], fb[c][0], acc[1][0]);
                const float nb0 = bs[kk * LD];
                if (wr) T[(kq + 2) * LD + r] = st[p].z;
                SB();
                MF(fa[c][1], fb[c][1], acc[1][1]);
                const float nb1 = bs[kk * LD + 32];
                if (wr) T[(kq + 3) * LD + r] = st[p].w;
                if (g == 11) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                SB();
                fa[c][0] = na0; fa[c][1] = na1; fb[c][0] = nb0; fb[c][1] = nb1;
            }
        }
        res = fa[0][0] + fb[1][1];
    } else {
        float4 fa[2][2], fb[2][2];
        for (int c = 0; c < 2; ++c) for (int i = 0; i < 2; ++i) { fa[c][i] = make_float4(1, 1, 1, 1); fb[c][i] = make_float4(1, 1, 1, 1); }
        for (int t = 0; t < iters; ++t) {
            const int cur = t & 1;
            const float* as = As[cur] + (wm * 64 + l31) * 36 + half * 4;
            const float* bs = Bs[cur] + (wn * 64 + l31) * 36 + half * 4;
#pragma unroll
            for (int qg = 0; qg < 4; ++qg) {
                const int c = qg & 1, nq = (qg + 2) & 3;
                float4 na[2], nb[2];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = sel(fa[c][0], e), a1 = sel(fa[c][1], e);
                    const float b0 = sel(fb[c][0], e), b1 = sel(fb[c][1], e);
                    MF(a0, b0, acc[0][0]);
                    if (e == 0) na[0] = *reinterpret_cast<const float4*>(as + nq * 8);
                    if (e == 1) nb[0] = *reinterpret_cast<const float4*>(bs + nq * 8);
                    SB();
                    MF(a0, b1, acc[0][1]);
                    if (e == 0) na[1] = *reinterpret_cast<const float4*>(as + 32 * 36 + nq * 8);
                    if (e == 1) nb[1] = *reinterpret_cast<const float4*>(bs + 32 * 36 + nq * 8);
                    SB();
                    MF(a1, b0, acc[1][0]);
                    if (e >= 2 && qg < 2) {
                        const int p = qg * 4 + (e - 2) * 2;
                        const int q = threadIdx.x + 256 * (p & 3);
                        float* T = (p < 4 ? As[cur ^ 1] : Bs[cur ^ 1]);
                        *reinterpret_cast<float4*>(T + (q >> 3) * 36 + ((q & 7) << 2)) = st[p];
                    }
                    SB();
                    MF(a1, b1, acc[1][1]);
                    if (e >= 2 && qg < 2) {
                        const int p = qg * 4 + (e - 2) * 2 + 1;
                        const int q = threadIdx.x + 256 * (p & 3);
                        float* T = (p < 4 ? As[cur ^ 1] : Bs[cur ^ 1]);
                        *reinterpret_cast<float4*>(T + (q >> 3) * 36 + ((q & 7) << 2)) = st[p];
                    }
                    if (qg == 2 && e == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    SB();
                }
                fa[c][0] = na[0]; fa[c][1] = na[1]; fb[c][0] = nb[0]; fb[c][1] = nb[1];
            }
        }
        res = fa[0][0].x + fb[1][1].y;
    }
    float s = res;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int W>
void run2(const char* name, float* out, int wg_per_cu) {
    const int iters = 2000, grid = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k2<MODE, W>), dim3(grid), dim3(256), 0, 0, out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k2<MODE, W>), dim3(grid), dim3(256), 0, 0, out, iters);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)grid * 4 * iters * 64 * 4096.0;
    printf("%-34s mode %d waves/eu %d wg/cu %d : %8.3f ms  %7.1f TFLOP/s\n", name, MODE, W, wg_per_cu, ms, fl / ms / 1e9);
}

int main() {
    float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
    run2<4, 1>("b32 spread 1 LDS op per MFMA", out, 1);
    run2<4, 2>("b32 spread 1 LDS op per MFMA", out, 2);
    run2<5, 1>("b128 layout, spread", out, 1);
    run2<5, 2>("b128 layout, spread", out, 1);
    run2<5, 2>("b128 layout, spread", out, 2);
    float* src; (void)hipMalloc(&src, 4096 * 4096 * 4); (void)hipMemset(src, 0, 4096 * 4096 * 4);
    run3<6, 1>("+8 global loads spread", out, src, 1);
    run3<6, 2>("+8 global loads spread", out, src, 2);
    run3<7, 1>("+8 global loads clumped", out, src, 1);
    run3<7, 2>("+8 global loads clumped", out, src, 2);
    return 0;
}
