// fx_metrics.hip — on-device evaluation metrics: binary logloss and AUC over all
// validation predictions, replacing the D->H copy + Python list.extend per batch + scikit-learn on
// float64 of BaseModel.evaluate (rank_model.py:369-381, metrics.py:49-51).
//
//   logloss = mean of -(y log p + (1-y) log(1-p)),  p = clip((double)pred, eps, 1-eps),
//             eps = DBL_EPSILON            (sklearn.metrics.log_loss on float64 input)
//   AUC     = Mann-Whitney U with average ranks for ties / (n_pos n_neg)
//             (= sklearn.metrics.roc_auc_score: the trapezoid under the ROC curve)
// AUC is computed in exact integer arithmetic: after a device radix sort of the predictions, every
// tie group [i0, i0+g) with gp positives adds gp * (2 i0 + g + 1) (= twice its positives' rank sum)
// to a 64-bit counter; integer atomics commute, so the result is deterministic and the host
// finishes with one float64 division.  n <= 2^26 keeps 3 n^2 inside 64 bits.
#include "fx_common.h"

#include <float.h>

#define FX_METRIC_BLOCKS 1024

__global__ __launch_bounds__(256) void k_metric_keys(const float* pred, const float* label, int64_t n,
                                                     uint32_t* key, uint32_t* val,
                                                     double* ll_partial) {
    __shared__ double red[256];
    double acc = 0.0;
    const double eps = DBL_EPSILON;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float pf = pred[i];
        uint32_t u = __float_as_uint(pf);
        u ^= (u >> 31) ? 0xFFFFFFFFu : 0x80000000u;          // order-preserving float -> uint
        key[i] = u;
        const float y = label[i];
        val[i] = y > 0.5f ? 1u : 0u;
        double p = (double)pf;
        p = p < eps ? eps : (p > 1.0 - eps ? 1.0 - eps : p);
        acc -= (double)y * log(p) + (1.0 - (double)y) * log(1.0 - p);   // sklearn's xlogy form
    }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ll_partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_metric_finish_ll(const double* ll_partial, int nb,
                                                          double* out, unsigned long long* cnt) {
    __shared__ double red[256];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) acc += ll_partial[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = red[0];
        cnt[0] = 0ull;   // twice the rank sum of the positives
        cnt[1] = 0ull;   // number of positives
    }
}

// one thread per sorted element; the head of a tie group walks its group
__global__ __launch_bounds__(256) void k_metric_ranks(const uint32_t* key, const uint32_t* val,
                                                      int64_t n, unsigned long long* cnt) {
    unsigned long long s2 = 0ull, np = 0ull;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const uint32_t k = key[i];
        if (i > 0 && key[i - 1] == k) continue;
        unsigned long long g = 0ull, gp = 0ull;
        for (int64_t j = i; j < n && key[j] == k; ++j) {
            ++g;
            gp += val[j];
        }
        s2 += gp * (2ull * (unsigned long long)i + g + 1ull);
        np += gp;
    }
    if (s2) atomicAdd(&cnt[0], s2);
    if (np) atomicAdd(&cnt[1], np);
}

static inline size_t fx_up(size_t x) { return (x + 255) / 256 * 256; }

extern "C" size_t fx_binary_metrics_workspace_bytes(int64_t n) {
    if (n <= 0) return 256;
    // key/value in, out and the sort's ping-pong scratch (fx_sort.hip: 4 passes over 32-bit keys)
    return 6 * fx_up((size_t)n * 4) + fx_up(FX_METRIC_BLOCKS * sizeof(double)) +
           fx_up(fx_sort_temp_bytes(n)) + 256;
}

extern "C" int fx_binary_metrics(const float* y_pred, const float* y_true, int64_t n,
                                 void* workspace, size_t workspace_bytes, double* out_logloss_sum,
                                 uint64_t* out_counts, fx_stream_t stream) {
    FX_CHECK_ARG(n >= 1 && n <= ((int64_t)1 << 26), "fx_binary_metrics: n=%lld not in [1, 2^26]",
                 (long long)n);
    FX_CHECK_ARG(y_pred && y_true && workspace && out_logloss_sum && out_counts,
                 "fx_binary_metrics: null pointer");
    const size_t sort_bytes = fx_sort_temp_bytes(n);
    const size_t arr = fx_up((size_t)n * 4), llb = fx_up(FX_METRIC_BLOCKS * sizeof(double));
    FX_CHECK_ARG(workspace_bytes >= 6 * arr + llb + fx_up(sort_bytes),
                 "fx_binary_metrics: workspace too small");
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* key_in = reinterpret_cast<uint32_t*>(w);
    uint32_t* val_in = reinterpret_cast<uint32_t*>(w + arr);
    uint32_t* key = reinterpret_cast<uint32_t*>(w + 2 * arr);
    uint32_t* val = reinterpret_cast<uint32_t*>(w + 3 * arr);
    uint32_t* key_tmp = reinterpret_cast<uint32_t*>(w + 4 * arr);
    uint32_t* val_tmp = reinterpret_cast<uint32_t*>(w + 5 * arr);
    double* llp = reinterpret_cast<double*>(w + 6 * arr);
    void* temp = w + 6 * arr + llb;
    hipStream_t s = fx_hip_stream(stream);
    hipLaunchKernelGGL(k_metric_keys, dim3(FX_METRIC_BLOCKS), dim3(256), 0, s, y_pred, y_true, n,
                       key_in, val_in, llp);
    hipLaunchKernelGGL(k_metric_finish_ll, dim3(1), dim3(256), 0, s, llp, (int)FX_METRIC_BLOCKS,
                       out_logloss_sum, reinterpret_cast<unsigned long long*>(out_counts));
    FX_CHECK_LAUNCH();
    const int rc = fx_sort_pairs_u32(key_in, val_in, key, val, key_tmp, val_tmp, n, 32u, temp, false,
                                     s);
    if (rc != FX_OK) return rc;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_metric_ranks, dim3((unsigned)blocks), dim3(256), 0, s, key, val, n,
                       reinterpret_cast<unsigned long long*>(out_counts));
    FX_CHECK_LAUNCH();
    return FX_OK;
}
