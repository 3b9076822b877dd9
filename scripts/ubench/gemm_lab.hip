// gemm_lab.hip — stand-alone bench / timeline / correctness harness around the PRODUCTION GEMM source
// (fuxictr_amd/csrc/fx_gemm.hip is #included with FX_GEMM_LAB, which adds per-workgroup timestamps).
// No torch: starts in a second on the GPU box.  Kernel switches are the library's own environment
// variables (FX_GEMM_TILE, FX_GEMM_TR, FX_GEMM_W64, FX_GEMM_PAIR ...), read once per process — run one
// process per configuration.
//   build:  hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -DFX_GEMM_LAB \
//               scripts/ubench/gemm_lab.hip -o scripts/ubench/gemm_lab
//   run:    gemm_lab [suite] [--trace] [--check]      suite = tower | cross | ksweep | all
#define FX_GEMM_LAB 1
#include "../../fuxictr_amd/csrc/fx_gemm.hip"
#include "../../fuxictr_amd/csrc/fx_gemm_x6.hip"     // (round 5: fx_gemm.hip dispatches into it; FX_GEMM_BF16X6=0 keeps the lab on the fp32 kernels)

#include <stdarg.h>

#include <math.h>

#include <algorithm>
#include <string>
#include <vector>

void fx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
}

#define HC(x)                                                                        \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(2);                                                                 \
        }                                                                            \
    } while (0)

__global__ void k_fill(float* p, int64_t n, uint32_t seed, float scale, float shift) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint32_t x = (uint32_t)i * 2654435761u + seed;
        x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
        p[i] = ((float)(x >> 8) * (1.0f / 16777216.0f) - 0.5f) * scale + shift;
    }
}

// reference: one thread per output, k-ordered fmaf chain (what the MFMA computes bit for bit when
// split_k == 1), then the same epilogue
__global__ void k_ref(GemmArgs a, int ta, int tb) {
    const int64_t total = a.M * a.N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / a.N, n = i - m * a.N;
        float acc = 0.f;
        for (int64_t k = 0; k < a.K; ++k) acc = fmaf(fx_a_at(a, ta, m, k), fx_b_at(a, tb, k, n), acc);
        a.C[m * a.ldc + n] = fx_epilogue(a.epi, acc, m, n);
    }
}

static float* dalloc(int64_t n, uint32_t seed, float scale = 2.f, float shift = 0.f) {
    float* p;
    HC(hipMalloc(&p, (size_t)std::max<int64_t>(n, 4) * 4));
    hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, p, n, seed, scale, shift);
    return p;
}

struct Case {
    std::string name;
    int ta, tb;
    int64_t M, N, K;
    int sk;
    bool bias, relu, mask, add, mul, zout, rowsum;
};

static bool g_trace = false, g_check = false;
static std::string g_tag;

static void analyse_trace(const std::vector<unsigned long long>& tr, int64_t nwg) {
    // words: 0 start, 1 first tile in LDS, 2 K loop done, 3 stores done, 4 HW_ID, 5 XCC_ID; 100 MHz
    unsigned long long t0 = ~0ull, t3 = 0;
    for (int64_t w = 0; w < nwg; ++w) {
        t0 = std::min(t0, tr[w * 8]);
        t3 = std::max(t3, tr[w * 8 + 3]);
    }
    auto stat = [&](const char* what, auto f) {
        std::vector<double> v;
        for (int64_t w = 0; w < nwg; ++w) v.push_back(f(w) * 0.01);
        std::sort(v.begin(), v.end());
        printf("    %-26s min %7.2f  p10 %7.2f  med %7.2f  p90 %7.2f  max %7.2f us\n", what, v[0],
               v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back());
    };
    printf("  trace: %lld workgroups, span %.2f us (first start -> last store done)\n", (long long)nwg,
           (t3 - t0) * 0.01);
    stat("start (after first WG)", [&](int64_t w) { return (double)(tr[w * 8] - t0); });
    stat("prologue (-> tile 0 in LDS)", [&](int64_t w) { return (double)(tr[w * 8 + 1] - tr[w * 8]); });
    stat("K loop", [&](int64_t w) { return (double)(tr[w * 8 + 2] - tr[w * 8 + 1]); });
    stat("epilogue (stores landed)", [&](int64_t w) { return (double)(tr[w * 8 + 3] - tr[w * 8 + 2]); });
    stat("end (before last WG)", [&](int64_t w) { return (double)(t3 - tr[w * 8 + 3]); });
    // per-XCD end times
    double xe[8] = {0}, xs[8];
    int xc[8] = {0};
    for (int x = 0; x < 8; ++x) xs[x] = 1e30;
    for (int64_t w = 0; w < nwg; ++w) {
        const int x = (int)(tr[w * 8 + 5] & 7);
        xe[x] = std::max(xe[x], (double)(tr[w * 8 + 3] - t0) * 0.01);
        xs[x] = std::min(xs[x], (double)(tr[w * 8] - t0) * 0.01);
        xc[x]++;
    }
    printf("    per XCD (wgs, first start, last end):");
    for (int x = 0; x < 8; ++x) printf(" [%d: %d %.1f %.1f]", x, xc[x], xs[x], xe[x]);
    printf("\n");
}

static void run_case(const Case& c, int reps = 20) {
    const int64_t M = c.M, N = c.N, K = c.K;
    float* A = dalloc(M * K, 1);
    float* B = dalloc(N * K, 2);
    float* C = dalloc(M * N, 3);
    float* Cref = g_check ? dalloc(M * N, 4) : nullptr;
    float* ws = dalloc((int64_t)std::max(c.sk, 1) * M * (N + N / 64 + 2) + 1024, 5);
    fx_gemm_epilogue e;
    memset(&e, 0, sizeof(e));
    if (c.bias) e.bias = dalloc(N, 6);
    if (c.relu) e.act = 1;
    if (c.mask) { e.mask = dalloc(M * N, 7); e.ldmask = N; }
    if (c.add) { e.add = dalloc(M * N, 8); e.ldadd = N; }
    if (c.mul) { e.mul = dalloc(M * N, 9); e.ldmul = N; }
    if (c.zout) { e.zout = dalloc(M * N, 10); e.ldz = N; }
    if (c.rowsum) e.rowsum = dalloc(M, 11);
    const int64_t lda = c.ta ? M : K, ldb = c.tb ? K : N;
    auto launch = [&]() {
        const int rc = fx_gemm_f32(c.ta, c.tb, M, N, K, A, lda, B, ldb, C, N, &e, c.sk, ws, nullptr);
        if (rc != FX_OK) { fprintf(stderr, "fx_gemm_f32 rc=%d\n", rc); exit(3); }
    };
    for (int i = 0; i < 3; ++i) launch();
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HC(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        HC(hipEventRecord(e1, 0));
        HC(hipEventSynchronize(e1));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms * 1000.f / reps);
    }
    const double tf = 2.0 * M * N * K / (best * 1e-6) / 1e12;
    printf("[%s] %-40s %8.2f us  %7.2f TF  %.3f\n", g_tag.c_str(), c.name.c_str(), best, tf, tf / 157.3);
    if (g_check) {
        GemmArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.A = A; ra.lda = lda; ra.B = B; ra.ldb = ldb; ra.C = Cref; ra.ldc = N;
        ra.M = M; ra.N = N; ra.K = K; ra.epi = e;
        ra.epi.zout = nullptr; ra.epi.rowsum = nullptr;
        hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, ra, c.ta, c.tb);
        launch();
        HC(hipDeviceSynchronize());
        std::vector<float> h1((size_t)M * N), h2((size_t)M * N);
        HC(hipMemcpy(h1.data(), C, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        HC(hipMemcpy(h2.data(), Cref, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0;
        int64_t nbad = 0;
        for (size_t i = 0; i < h1.size(); ++i) {
            const double d = fabs((double)h1[i] - (double)h2[i]);
            md = std::max(md, d);
            mx = std::max(mx, fabs((double)h2[i]));
            if (memcmp(&h1[i], &h2[i], 4) != 0) ++nbad;
        }
        printf("  check: max |d| %.3e (max |ref| %.3e), %lld of %lld elements not bit-identical%s\n", md,
               mx, (long long)nbad, (long long)h1.size(),
               (c.sk == 1 && nbad) ? "   <-- MISMATCH" : (md > 1e-3 * mx ? "   <-- MISMATCH" : ""));
    }
    if (g_trace) {
        const int64_t words = (int64_t)1 << 20;
        unsigned long long* tr;
        HC(hipMalloc(&tr, words * 8));
        HC(hipMemset(tr, 0, words * 8));
        fx_gemm_lab_trace = tr;
        launch();
        HC(hipDeviceSynchronize());
        fx_gemm_lab_trace = nullptr;
        std::vector<unsigned long long> h(words);
        HC(hipMemcpy(h.data(), tr, words * 8, hipMemcpyDeviceToHost));
        int64_t nwg = 0;
        while (nwg < words / 8 && h[nwg * 8] != 0) ++nwg;
        if (nwg > 0) analyse_trace(h, nwg);
        HC(hipFree(tr));
    }
    HC(hipFree(A)); HC(hipFree(B)); HC(hipFree(C)); HC(hipFree(ws));
    if (Cref) HC(hipFree(Cref));
}

static void run_pair(const char* name, int64_t M, int64_t N, int64_t K, int sk, bool mask, bool add,
                     int reps = 20) {
    // dW[N,K] = dz^T x (split-K, rowsum) + dX[M,K] = dz W (mask / add), one fx_gemm_f32_batch call
    float* dz = dalloc(M * N, 1);
    float* x = dalloc(M * K, 2);
    float* W = dalloc(N * K, 3);
    float* dW = dalloc(N * K, 4);
    float* dx = dalloc(M * K, 5);
    float* ws = dalloc((int64_t)sk * N * (K + K / 64 + 2) + 1024, 6);
    fx_gemm_epilogue e1, e2;
    memset(&e1, 0, sizeof(e1));
    memset(&e2, 0, sizeof(e2));
    e1.rowsum = dalloc(N, 7);
    if (mask) { e2.mask = dalloc(M * K, 8); e2.ldmask = K; }
    if (add) { e2.add = dalloc(M * K, 9); e2.ldadd = K; }
    fx_gemm_problem p[2];
    memset(p, 0, sizeof(p));
    p[0].transa = 1; p[0].transb = 0; p[0].M = N; p[0].N = K; p[0].K = M;
    p[0].A = dz; p[0].lda = N; p[0].B = x; p[0].ldb = K; p[0].C = dW; p[0].ldc = K;
    p[0].epilogue = &e1; p[0].split_k = sk; p[0].workspace = ws;
    p[1].transa = 0; p[1].transb = 0; p[1].M = M; p[1].N = K; p[1].K = N;
    p[1].A = dz; p[1].lda = N; p[1].B = W; p[1].ldb = K; p[1].C = dx; p[1].ldc = K;
    p[1].epilogue = &e2; p[1].split_k = 1; p[1].workspace = nullptr;
    auto launch = [&]() {
        if (fx_gemm_f32_batch(p, 2, nullptr) != FX_OK) { fprintf(stderr, "batch failed\n"); exit(3); }
    };
    for (int i = 0; i < 3; ++i) launch();
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1v;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1v));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HC(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        HC(hipEventRecord(e1v, 0));
        HC(hipEventSynchronize(e1v));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1v));
        best = std::min(best, ms * 1000.f / reps);
    }
    const double tf = 4.0 * M * N * K / (best * 1e-6) / 1e12;
    printf("[%s] %-40s %8.2f us  %7.2f TF  %.3f  (incl. slab reduce)\n", g_tag.c_str(), name, best, tf,
           tf / 157.3);
    if (g_check) {
        float* r1 = dalloc(N * K, 14);
        float* r2 = dalloc(M * K, 15);
        GemmArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.A = dz; ra.lda = N; ra.B = x; ra.ldb = K; ra.C = r1; ra.ldc = K; ra.M = N; ra.N = K; ra.K = M;
        hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, ra, 1, 0);
        memset(&ra, 0, sizeof(ra));
        ra.A = dz; ra.lda = N; ra.B = W; ra.ldb = K; ra.C = r2; ra.ldc = K; ra.M = M; ra.N = K; ra.K = N;
        ra.epi = e2;
        hipLaunchKernelGGL(k_ref, dim3(4096), dim3(256), 0, 0, ra, 0, 0);
        launch();
        HC(hipDeviceSynchronize());
        auto cmp = [&](const char* what, float* got, float* ref, int64_t n, bool exact) {
            std::vector<float> h1((size_t)n), h2((size_t)n);
            HC(hipMemcpy(h1.data(), got, (size_t)n * 4, hipMemcpyDeviceToHost));
            HC(hipMemcpy(h2.data(), ref, (size_t)n * 4, hipMemcpyDeviceToHost));
            double md = 0, mx = 0;
            int64_t nbad = 0;
            for (size_t i = 0; i < h1.size(); ++i) {
                md = std::max(md, fabs((double)h1[i] - (double)h2[i]));
                mx = std::max(mx, fabs((double)h2[i]));
                if (memcmp(&h1[i], &h2[i], 4) != 0) ++nbad;
            }
            printf("  check %s: max |d| %.3e (max |ref| %.3e), %lld not bit-identical%s\n", what, md, mx,
                   (long long)nbad, ((exact && nbad) || md > 2e-5 * mx * sqrt((double)M)) ? "   <-- MISMATCH" : "");
        };
        cmp("dW", dW, r1, N * K, false);
        cmp("dX", dx, r2, M * K, true);
        // bias gradient
        std::vector<float> hb((size_t)N), hz((size_t)M * N);
        HC(hipMemcpy(hb.data(), e1.rowsum, (size_t)N * 4, hipMemcpyDeviceToHost));
        HC(hipMemcpy(hz.data(), dz, (size_t)M * N * 4, hipMemcpyDeviceToHost));
        double mdb = 0;
        for (int64_t n2 = 0; n2 < N; ++n2) {
            double sacc = 0;
            for (int64_t m2 = 0; m2 < M; ++m2) sacc += hz[(size_t)m2 * N + n2];
            mdb = std::max(mdb, fabs(sacc - (double)hb[n2]));
        }
        printf("  check db: max |d| %.3e%s\n", mdb, mdb > 1e-2 ? "   <-- MISMATCH" : "");
        HC(hipFree(r1)); HC(hipFree(r2));
    }
    HC(hipFree(dz)); HC(hipFree(x)); HC(hipFree(W)); HC(hipFree(dW)); HC(hipFree(dx)); HC(hipFree(ws));
}

// several problems through fx_gemm_f32_batch.  spec: "k:M,N,K;k:M,N,K;..." (layer sizes: batch M, out N,
// in K) with kind k = f forward (bias + relu), c cross forward (bias + zout + mul + add), x input gradient
// (relu mask), a input gradient (residual add), w weight gradient (K split allowed, fused bias gradient)
static void run_mix(const char* name, const char* spec, int sk, int reps = 20) {
    fx_gemm_problem p[4];
    fx_gemm_epilogue e[4];
    memset(p, 0, sizeof(p));
    memset(e, 0, sizeof(e));
    double flops = 0;
    int n = 0;
    const char* q = spec;
    while (*q && n < 4) {
        const char kind = *q;
        long long M = 0, N = 0, K = 0;
        sscanf(q + 2, "%lld,%lld,%lld", &M, &N, &K);
        const int i = n++;
        if (kind == 'f' || kind == 'c') {
            p[i].transa = 0; p[i].transb = 1; p[i].M = M; p[i].N = N; p[i].K = K;
            p[i].A = dalloc(M * K, 31 + i); p[i].lda = K; p[i].B = dalloc(N * K, 41 + i); p[i].ldb = K;
            p[i].C = dalloc(M * N, 51 + i); p[i].ldc = N;
            e[i].bias = dalloc(N, 61 + i);
            if (kind == 'f') e[i].act = 1;
            else {
                e[i].zout = dalloc(M * N, 81 + i); e[i].ldz = N;
                e[i].mul = dalloc(M * N, 91 + i); e[i].ldmul = N;
                e[i].add = dalloc(M * N, 101 + i); e[i].ldadd = N;
            }
            p[i].split_k = 1;
        } else if (kind == 'x' || kind == 'a') {
            p[i].transa = 0; p[i].transb = 0; p[i].M = M; p[i].N = K; p[i].K = N;
            p[i].A = dalloc(M * N, 31 + i); p[i].lda = N; p[i].B = dalloc(N * K, 41 + i); p[i].ldb = K;
            p[i].C = dalloc(M * K, 51 + i); p[i].ldc = K;
            if (kind == 'x') { e[i].mask = dalloc(M * K, 61 + i); e[i].ldmask = K; }
            else { e[i].add = dalloc(M * K, 61 + i); e[i].ldadd = K; }
            p[i].split_k = 1;
        } else {
            p[i].transa = 1; p[i].transb = 0; p[i].M = N; p[i].N = K; p[i].K = M;
            p[i].A = dalloc(M * N, 31 + i); p[i].lda = N; p[i].B = dalloc(M * K, 41 + i); p[i].ldb = K;
            p[i].C = dalloc(N * K, 51 + i); p[i].ldc = K;
            e[i].rowsum = dalloc(N, 61 + i);
            p[i].split_k = sk; p[i].workspace = dalloc((int64_t)sk * N * (K + K / 64 + 2) + 1024, 71 + i);
        }
        p[i].epilogue = &e[i];
        flops += 2.0 * M * N * K;
        while (*q && *q != ';') ++q;
        if (*q == ';') ++q;
    }
    auto launch = [&]() {
        if (fx_gemm_f32_batch(p, n, nullptr) != FX_OK) { fprintf(stderr, "batch failed\n"); exit(3); }
    };
    for (int i = 0; i < 3; ++i) launch();
    HC(hipDeviceSynchronize());
    hipEvent_t e0, e1v;
    HC(hipEventCreate(&e0));
    HC(hipEventCreate(&e1v));
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HC(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch();
        HC(hipEventRecord(e1v, 0));
        HC(hipEventSynchronize(e1v));
        float ms;
        HC(hipEventElapsedTime(&ms, e0, e1v));
        best = std::min(best, ms * 1000.f / reps);
    }
    const double tf = flops / (best * 1e-6) / 1e12;
    printf("[%s] %-40s %8.2f us  %7.2f TF  %.3f\n", g_tag.c_str(), name, best, tf, tf / 157.3);
    if (g_trace) {
        const int64_t words = (int64_t)1 << 20;
        unsigned long long* tr;
        HC(hipMalloc(&tr, words * 8));
        HC(hipMemset(tr, 0, words * 8));
        fx_gemm_lab_trace = tr;
        launch();
        HC(hipDeviceSynchronize());
        fx_gemm_lab_trace = nullptr;
        std::vector<unsigned long long> h(words);
        HC(hipMemcpy(h.data(), tr, words * 8, hipMemcpyDeviceToHost));
        int64_t nwg = 0;
        while (nwg < words / 8 && h[nwg * 8] != 0) ++nwg;
        if (nwg > 0) {
            analyse_trace(h, nwg);
            // first / second half of the grid separately (problem order inside a multi launch)
            std::vector<unsigned long long> h1(h.begin(), h.begin() + (nwg / 2) * 8);
            std::vector<unsigned long long> h2(h.begin() + (nwg / 2) * 8, h.begin() + nwg * 8);
            printf("  -- first half of the grid\n");
            analyse_trace(h1, nwg / 2);
            printf("  -- second half of the grid\n");
            analyse_trace(h2, nwg - nwg / 2);
        }
        HC(hipFree(tr));
    }
}

int main(int argc, char** argv) {
    std::string suite = "tower";
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--trace")) g_trace = true;
        else if (!strcmp(argv[i], "--check")) g_check = true;
        else suite = argv[i];
    }
    auto env = [](const char* k, const char* d) { const char* e = getenv(k); return std::string(e ? e : d); };
    g_tag = "tile=" + env("FX_GEMM_TILE", "auto") + " tr=" + env("FX_GEMM_TR", "1") + env("FX_LAB_TAG", "");
    const int64_t B = 4096;
    {   // DVFS / power-state warm-up: ~60 ms of GEMMs before anything is timed (the first cases of a cold
        // process measured ~12 % slow in visit a)
        float* wa = dalloc(4096 * 1024, 21);
        float* wb = dalloc(1024 * 1024, 22);
        float* wc = dalloc(4096 * 1024, 23);
        for (int i = 0; i < 600; ++i)
            fx_gemm_f32(0, 1, 4096, 1024, 1024, wa, 1024, wb, 1024, wc, 1024, nullptr, 1, nullptr, nullptr);
        HC(hipDeviceSynchronize());
        HC(hipFree(wa)); HC(hipFree(wb)); HC(hipFree(wc));
    }
    if (suite == "two") {   // two 128x128 workgroups per CU: how well do they share the matrix pipes?
        run_case({"fwd 8192x1024x1024 bias+relu", 0, 1, 8192, 1024, 1024, 1, true, true, false, false, false, false, false});
        run_case({"fwd 8192x1024x1024 plain", 0, 1, 8192, 1024, 1024, 1, false, false, false, false, false, false, false});
        run_case({"fwd 4096x1024x1024 plain", 0, 1, B, 1024, 1024, 1, false, false, false, false, false, false, false});
        run_case({"dW 1024x1024x4096 sk8 rowsum", 1, 0, 1024, 1024, B, 8, false, false, false, false, false, false, true});
        run_case({"dX 4096x1024x1024 plain", 0, 0, B, 1024, 1024, 1, false, false, false, false, false, false, false});
        run_case({"dX 4096x1024x1024 relu mask", 0, 0, B, 1024, 1024, 1, false, false, true, false, false, false, false});
    }
    if (suite == "mix") {   // where does a multi-problem launch lose time?
        run_mix("multi: fwd alone (n = 1)", "f:4096,1024,1024", 1);
        run_mix("multi: fwd + fwd", "f:4096,1024,1024;f:4096,1024,1024", 1);
        run_mix("multi: dX + dX", "x:4096,1024,1024;x:4096,1024,1024", 1);
        run_mix("multi: dW alone (cap 4)", "w:4096,1024,1024", 4);
        run_mix("multi: dW + dW (cap 4)", "w:4096,1024,1024;w:4096,1024,1024", 4);
        run_mix("multi: dW + dX (cap 4)", "w:4096,1024,1024;x:4096,1024,1024", 4);
        run_mix("multi: fwd + dX", "f:4096,1024,1024;x:4096,1024,1024", 1);
    }
    if (suite == "dcn") {   // DCNv2 parallel: cross layer + deep layer of one depth
        run_mix("cross fwd alone", "c:4096,624,624", 1);
        run_mix("deep fwd 1024x1024 alone", "f:4096,1024,1024", 1);
        run_mix("deep fwd 1024x624 alone", "f:4096,1024,624", 1);
        run_mix("cross fwd + deep fwd 1024x1024", "c:4096,624,624;f:4096,1024,1024", 1);
        run_mix("cross fwd + deep fwd 1024x624", "c:4096,624,624;f:4096,1024,624", 1);
        run_mix("cross pair alone (dW + dX add)", "w:4096,624,624;a:4096,624,624", 8);
        run_mix("deep pair 1024x1024 alone", "w:4096,1024,1024;x:4096,1024,1024", 8);
        run_mix("deep pair 1024x624 alone", "w:4096,1024,624;a:4096,1024,624", 8);
        run_mix("cross pair + deep pair 1024x1024", "w:4096,624,624;a:4096,624,624;w:4096,1024,1024;x:4096,1024,1024", 8);
        run_mix("cross pair + deep pair 1024x624", "w:4096,624,624;a:4096,624,624;w:4096,1024,624;a:4096,1024,624", 8);
    }
    if (suite == "first") {   // the tower's first layer (K = N = 624 = 39 x 16): as it is, and padded to 640
        run_case({"fwd 4096x1024x624 bias+relu", 0, 1, B, 1024, 624, 1, true, true, false, false, false, false, false});
        run_case({"fwd 4096x1024x640 bias+relu", 0, 1, B, 1024, 640, 1, true, true, false, false, false, false, false});
        run_mix("first-layer pair 624 (dW + dX)", "w:4096,1024,624;x:4096,1024,624", 8);
        run_mix("first-layer pair 640 (dW + dX)", "w:4096,1024,640;x:4096,1024,640", 8);
    }
    if (suite == "pairs") {
        run_pair("pair 4096x1024x1024 (dW + dX mask)", B, 1024, 1024, 8, true, false);
        run_pair("pair 4096x1024x624 (dW + dX)", B, 1024, 624, 8, false, false);
        run_pair("cross pair 4096x624x624 (dW + dX add)", B, 624, 624, 8, false, true);
        run_pair("pair 4096x512x1024 (dW + dX mask)", B, 512, 1024, 8, true, false);
        run_pair("pair 4096x256x512 (dW + dX mask)", B, 256, 512, 8, true, false);
        run_pair("pair 4096x1024x368 (dW + dX)", B, 1024, 368, 8, false, false);
    }
    if (suite == "tower" || suite == "all") {
        run_case({"fwd 4096x1024x1024 bias+relu", 0, 1, B, 1024, 1024, 1, true, true, false, false, false, false, false});
        run_case({"fwd 4096x1024x624 bias+relu", 0, 1, B, 1024, 624, 1, true, true, false, false, false, false, false});
        run_case({"dX 4096x1024x1024 relu mask", 0, 0, B, 1024, 1024, 1, false, false, true, false, false, false, false});
        run_case({"dW 1024x1024x4096 sk4 rowsum", 1, 0, 1024, 1024, B, 4, false, false, false, false, false, false, true});
        run_pair("pair 4096x1024x1024 (dW sk4 + dX mask)", B, 1024, 1024, 4, true, false);
        run_pair("pair 4096x1024x624 (dW sk4 + dX)", B, 1024, 624, 4, false, false);
    }
    if (suite == "cross" || suite == "all") {
        run_case({"cross fwd 4096x624x624 bias+zout+mul+add", 0, 1, B, 624, 624, 1, true, false, false, true, true, true, false});
        run_case({"cross dX 4096x624x624 add", 0, 0, B, 624, 624, 1, false, false, false, true, false, false, false});
        run_case({"cross dW 624x624x4096 sk4 rowsum", 1, 0, 624, 624, B, 4, false, false, false, false, false, false, true});
        run_pair("cross pair 4096x624x624 (dW sk4 + dX add)", B, 624, 624, 4, false, true);
    }
    if (suite == "ksweep" || suite == "all") {
        for (int64_t K : {32, 64, 256, 512, 1024, 2048, 4096}) {
            char nm[64];
            snprintf(nm, sizeof(nm), "fwd 4096x1024 K=%lld bias+relu", (long long)K);
            run_case({nm, 0, 1, B, 1024, K, 1, true, true, false, false, false, false, false});
        }
        for (int64_t K : {32, 1024}) {
            char nm[64];
            snprintf(nm, sizeof(nm), "fwd 4096x1024 K=%lld plain", (long long)K);
            run_case({nm, 0, 1, B, 1024, K, 1, false, false, false, false, false, false, false});
        }
    }
    if (suite == "odd") {   // ragged shapes: correctness of the masked paths (use with --check)
        run_case({"fwd 1000x520x136 bias+relu", 0, 1, 1000, 520, 136, 1, true, true, false, false, false, false, false});
        run_case({"dX 333x260x72 mask+add", 0, 0, 333, 260, 72, 1, false, false, true, true, false, false, false});
        run_case({"dW 260x136x1000 sk3 rowsum", 1, 0, 260, 136, 1000, 3, false, false, false, false, false, false, true});
        run_case({"TN 200x68x40", 1, 1, 200, 68, 40, 1, true, false, false, false, true, true, false});
    }
    return 0;
}
