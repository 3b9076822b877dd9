// What does one kernel NODE of a replayed hipGraph cost on MI355X, by what the kernel does?
// VERDICT r3 #10 / housekeeping 9: the one-workgroup kernels of the captured training step (k_clip_coef,
// k_splitk_reduce_wide, the head GEMMs ...) show 4.4 - 4.8 us each in rocprofv3's kernel trace while
// MI355X_MICROARCH.md prices a dependent kernel boundary at 1.1 - 1.9 us.  This bench replays graphs of
// N dependent nodes of one kind between an event pair and prints us per node.
// build: hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ void k_empty() {}

__global__ void k_empty_args(const float* a, float* b, long n, int m) {}

struct Big { void* p[64]; long n[64]; };              // 1 KB by-value argument (pointer lists of the mt_* kernels)
__global__ void k_big_args(Big big, float* out) {
    if (big.n[threadIdx.x & 63] == -12345) out[0] = 1.f;
}

// one dependent global round trip: read a word, write it back
__global__ void k_one_load(const float* __restrict__ in, float* __restrict__ out) {
    out[threadIdx.x] = in[threadIdx.x] + 1.f;
}

// a chain of `depth` dependent loads (pointer chase through a small table: L2 hits after the first replay)
__global__ void k_chain(const int* __restrict__ next, int* __restrict__ out, int depth) {
    int i = threadIdx.x;
    for (int d = 0; d < depth; ++d) i = next[i];
    out[threadIdx.x] = i;
}

// the shape of k_clip_coef: one workgroup sums a few hundred partials in fp64 and writes 3 scalars
__global__ void k_reduce_small(const float* __restrict__ parts, int n, float* __restrict__ scal) {
    __shared__ double sh[256];
    double s = 0;
    for (int i = threadIdx.x; i < n; i += 256) s += (double)parts[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float nrm = (float)sqrt(sh[0]);
        scal[10] = nrm;
        scal[9] = nrm > scal[11] ? scal[11] / (nrm + 1e-6f) : 1.f;
    }
}

// a streaming kernel of `mb` MB (what a real neighbour looks like): read + write
__global__ void k_stream(const float4* __restrict__ in, float4* __restrict__ out, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) { float4 v = in[i]; v.x += 1.f; out[i] = v; }
}

template <typename F>
static float time_graph(hipStream_t s, int nodes, int reps, F launch_one) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nodes; ++i) launch_one(i);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 1e3f * ms / reps / nodes;
}

template <typename F>
static float time_eager(hipStream_t s, int nodes, int reps, F launch_one) {
    for (int i = 0; i < nodes; ++i) launch_one(i);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < reps; ++r) for (int i = 0; i < nodes; ++i) launch_one(i);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return 1e3f * ms / reps / nodes;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *a, *b; int *nx, *no; float* scal; float4 *sa, *sb;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMalloc(&scal, 256));
    CK(hipMalloc(&nx, 1 << 16)); CK(hipMalloc(&no, 1 << 16));
    const long big_n = (16l << 20) / 16;
    CK(hipMalloc(&sa, 16 << 20)); CK(hipMalloc(&sb, 16 << 20));
    CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(scal, 0, 256)); CK(hipMemset(sa, 0, 16 << 20));
    std::vector<int> h(1 << 14);
    for (int i = 0; i < (1 << 14); ++i) h[i] = (i * 7919 + 13) & ((1 << 14) - 1);
    CK(hipMemcpy(nx, h.data(), 1 << 16, hipMemcpyHostToDevice));
    Big big; for (int i = 0; i < 64; ++i) { big.p[i] = a; big.n[i] = i; }
    const int N = 200, R = 20;
    struct Row { const char* name; float graph, eager; };
    std::vector<Row> rows;
#define RUN(label, body) { auto f = [&](int i) { body; }; rows.push_back({label, time_graph(s, N, R, f), time_eager(s, N, R, f)}); }
    RUN("empty <<<1,64>>> no args", hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s));
    RUN("empty <<<1,256>>> 4 args", hipLaunchKernelGGL(k_empty_args, dim3(1), dim3(256), 0, s, a, b, 5l, 3));
    RUN("empty <<<256,256>>> 4 args", hipLaunchKernelGGL(k_empty_args, dim3(256), dim3(256), 0, s, a, b, 5l, 3));
    RUN("empty <<<1024,256>>> 4 args", hipLaunchKernelGGL(k_empty_args, dim3(1024), dim3(256), 0, s, a, b, 5l, 3));
    RUN("1 KB by-value args <<<1,64>>>", hipLaunchKernelGGL(k_big_args, dim3(1), dim3(64), 0, s, big, b));
    RUN("one load+store <<<1,64>>>", hipLaunchKernelGGL(k_one_load, dim3(1), dim3(64), 0, s, a, b));
    RUN("one load+store <<<256,256>>>", hipLaunchKernelGGL(k_one_load, dim3(256), dim3(256), 0, s, a, b));
    RUN("chain of 2 loads <<<1,64>>>", hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, nx, no, 2));
    RUN("chain of 4 loads <<<1,64>>>", hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, nx, no, 4));
    RUN("chain of 8 loads <<<1,64>>>", hipLaunchKernelGGL(k_chain, dim3(1), dim3(64), 0, s, nx, no, 8));
    RUN("fp64 sum of 768 partials <<<1,256>>> (k_clip_coef shape)", hipLaunchKernelGGL(k_reduce_small, dim3(1), dim3(256), 0, s, a, 768, scal));
    RUN("stream 16 MB in + 16 MB out <<<2048,256>>>", hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, sa, sb, big_n));
    // a tiny node BETWEEN two streaming kernels: marginal cost of the tiny node in a realistic neighbourhood
    {
        auto pairf = [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, sa, sb, big_n);
        };
        auto trio = [&](int i) {
            hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, s, sa, sb, big_n);
            hipLaunchKernelGGL(k_reduce_small, dim3(1), dim3(256), 0, s, a, 768, scal);
        };
        float t2 = time_graph(s, 100, R, pairf), t3 = time_graph(s, 100, R, trio);
        printf("%-62s %7.2f us (stream alone %.2f, stream + tiny %.2f per pair)\n",
               "marginal cost of the k_clip_coef-shaped node after a stream kernel", t3 - t2, t2, t3);
    }
    printf("%-62s %10s %10s\n", "kernel node (200 dependent nodes per graph, 20 replays)", "graph us", "eager us");
    for (auto& r : rows) printf("%-62s %10.2f %10.2f\n", r.name, r.graph, r.eager);
    return 0;
}
