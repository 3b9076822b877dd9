// row_record.hip — micro-benchmark behind DESIGN.md section 9: what would the [p | m | v | last] row record buy?
// The exact-mode catch-up and the sparse Adam update touch, per unique row, p, m, v (64 B each) and last_step
// (4 B) of the D = 16 table plus the same four arrays of the D = 1 table: eight scattered accesses into multi-GB
// arrays.  A 256-byte record per row (p 16 | m 16 | v 16 | last, p1, m1, v1 | pad) would make it one.
// Both layouts below do the same read-modify-write of 25 K random rows out of 33.76 M (one lane quad per row,
// every load issued before the first use), timed with HIP events over 200 launches on fresh row sets.
//   hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/row_record scripts/ubench/row_record.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_split(float* p, float* m, float* v, int* last, float* p1, float* m1, float* v1,
                                               int* last1, const uint32_t* rows, int n, int step) {
    const int sub = threadIdx.x & 3;
    for (int u = blockIdx.x * 64 + (threadIdx.x >> 2); u < n; u += gridDim.x * 64) {
        const int64_t r = rows[u];
        float4 a = *reinterpret_cast<float4*>(p + r * 16 + sub * 4);
        float4 b = *reinterpret_cast<float4*>(m + r * 16 + sub * 4);
        float4 c = *reinterpret_cast<float4*>(v + r * 16 + sub * 4);
        const int l = last[r], l1 = last1[r];
        float a1 = 0.f, b1 = 0.f, c1 = 0.f;
        if (sub == 0) { a1 = p1[r]; b1 = m1[r]; c1 = v1[r]; }
        const float k = (float)(step - l) * 1e-6f + (float)(step - l1) * 1e-7f;
        b.x *= 0.9f; b.y *= 0.9f; b.z *= 0.9f; b.w *= 0.9f;
        c.x *= 0.999f; c.y *= 0.999f; c.z *= 0.999f; c.w *= 0.999f;
        a.x -= k * b.x; a.y -= k * b.y; a.z -= k * b.z; a.w -= k * b.w;
        *reinterpret_cast<float4*>(p + r * 16 + sub * 4) = a;
        *reinterpret_cast<float4*>(m + r * 16 + sub * 4) = b;
        *reinterpret_cast<float4*>(v + r * 16 + sub * 4) = c;
        if (sub == 0) { p1[r] = a1 - k * b1; m1[r] = b1 * 0.9f; v1[r] = c1 * 0.999f; last[r] = step; last1[r] = step; }
    }
}

// record: 64 floats per row: [p 0..15 | m 16..31 | v 32..47 | last, p1, m1, v1, pad...]
__global__ __launch_bounds__(256) void k_record(float* rec, const uint32_t* rows, int n, int step) {
    const int sub = threadIdx.x & 3;
    for (int u = blockIdx.x * 64 + (threadIdx.x >> 2); u < n; u += gridDim.x * 64) {
        float* q = rec + (int64_t)rows[u] * 64;
        float4 a = *reinterpret_cast<float4*>(q + sub * 4);
        float4 b = *reinterpret_cast<float4*>(q + 16 + sub * 4);
        float4 c = *reinterpret_cast<float4*>(q + 32 + sub * 4);
        float4 t = *reinterpret_cast<float4*>(q + 48);            // (the four lanes read the same 16 bytes)
        const int l = __float_as_int(t.x);
        const float k = (float)(step - l) * 1.1e-6f;
        b.x *= 0.9f; b.y *= 0.9f; b.z *= 0.9f; b.w *= 0.9f;
        c.x *= 0.999f; c.y *= 0.999f; c.z *= 0.999f; c.w *= 0.999f;
        a.x -= k * b.x; a.y -= k * b.y; a.z -= k * b.z; a.w -= k * b.w;
        *reinterpret_cast<float4*>(q + sub * 4) = a;
        *reinterpret_cast<float4*>(q + 16 + sub * 4) = b;
        *reinterpret_cast<float4*>(q + 32 + sub * 4) = c;
        if (sub == 0) {
            t.x = __int_as_float(step); t.y -= k * t.z; t.z *= 0.9f; t.w *= 0.999f;
            *reinterpret_cast<float4*>(q + 48) = t;
        }
    }
}

// read-only variants: what a gather (p only) costs in both layouts, one quad per lookup
__global__ __launch_bounds__(256) void k_gather_split(const float* p, const float* p1, const uint32_t* rows, int n, float* out) {
    const int sub = threadIdx.x & 3;
    for (int u = blockIdx.x * 64 + (threadIdx.x >> 2); u < n; u += gridDim.x * 64) {
        const int64_t r = rows[u];
        float4 a = *reinterpret_cast<const float4*>(p + r * 16 + sub * 4);
        if (sub == 0) a.x += p1[r];
        *reinterpret_cast<float4*>(out + (int64_t)u * 16 + sub * 4) = a;
    }
}
__global__ __launch_bounds__(256) void k_gather_record(const float* rec, const uint32_t* rows, int n, float* out) {
    const int sub = threadIdx.x & 3;
    for (int u = blockIdx.x * 64 + (threadIdx.x >> 2); u < n; u += gridDim.x * 64) {
        const float* q = rec + (int64_t)rows[u] * 64;
        float4 a = *reinterpret_cast<const float4*>(q + sub * 4);
        if (sub == 0) a.x += q[49];
        *reinterpret_cast<float4*>(out + (int64_t)u * 16 + sub * 4) = a;
    }
}

int main(int argc, char** argv) {
    const int64_t R = 33762603;
    const int sets = 64;
    const int n_rows[2] = {25000, 106496};      // unique rows per step (power-law) / lookups per step
    float *p, *m, *v, *p1, *m1, *v1, *rec, *out;
    int *last, *last1;
    CHECK(hipMalloc(&p, R * 64)); CHECK(hipMalloc(&m, R * 64)); CHECK(hipMalloc(&v, R * 64));
    CHECK(hipMalloc(&p1, R * 4)); CHECK(hipMalloc(&m1, R * 4)); CHECK(hipMalloc(&v1, R * 4));
    CHECK(hipMalloc(&last, R * 4)); CHECK(hipMalloc(&last1, R * 4));
    CHECK(hipMalloc(&rec, R * 256));
    CHECK(hipMalloc(&out, (int64_t)n_rows[1] * 64));
    CHECK(hipMemset(p, 0, R * 64)); CHECK(hipMemset(m, 0, R * 64)); CHECK(hipMemset(v, 0, R * 64));
    CHECK(hipMemset(p1, 0, R * 4)); CHECK(hipMemset(m1, 0, R * 4)); CHECK(hipMemset(v1, 0, R * 4));
    CHECK(hipMemset(last, 0, R * 4)); CHECK(hipMemset(last1, 0, R * 4)); CHECK(hipMemset(rec, 0, R * 256));
    for (int which = 0; which < 2; ++which) {
        const int n = n_rows[which];
        std::vector<uint32_t> h((size_t)sets * n);
        uint64_t s = 88172645463325252ull + which;
        for (auto& x : h) {
            s ^= s << 13; s ^= s >> 7; s ^= s << 17;
            // unique rows of a step: uniform over the table (the de-dup removed the hot duplicates; the pessimistic case);
            // lookups of a step: power-law (card * u^3), duplicates included, as the bench's ids
            const double u = (double)(s >> 11) / 9007199254740992.0;
            x = (uint32_t)((double)(R - 1) * (which == 0 ? u : u * u * u));
        }
        uint32_t* rows;
        CHECK(hipMalloc(&rows, h.size() * 4));
        CHECK(hipMemcpy(rows, h.data(), h.size() * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int blocks = (n + 63) / 64;
        for (int mode = 0; mode < 2; ++mode) {
            float best = 1e30f, tot = 0.f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0));
                for (int i = 0; i < sets; ++i) {
                    const uint32_t* rr = rows + (size_t)i * n;
                    if (which == 0) {
                        if (mode == 0) hipLaunchKernelGGL(k_split, dim3(blocks), dim3(256), 0, 0, p, m, v, last, p1, m1, v1, last1, rr, n, 100 + i);
                        else hipLaunchKernelGGL(k_record, dim3(blocks), dim3(256), 0, 0, rec, rr, n, 100 + i);
                    } else {
                        if (mode == 0) hipLaunchKernelGGL(k_gather_split, dim3(blocks), dim3(256), 0, 0, p, p1, rr, n, out);
                        else hipLaunchKernelGGL(k_gather_record, dim3(blocks), dim3(256), 0, 0, rec, rr, n, out);
                    }
                }
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                best = ms < best ? ms : best;
                tot += ms;
            }
            printf("%-34s %-22s %7d rows: %6.2f us per launch (best of 4 x %d back-to-back launches on distinct row sets)\n",
                   which == 0 ? "read-modify-write p, m, v, last (+D=1)" : "gather p (+D=1) of every lookup",
                   mode == 0 ? "8 / 2 separate arrays" : "one 256-byte record", n, 1e3f * best / sets, sets);
        }
        CHECK(hipFree(rows));
    }
    return 0;
}
