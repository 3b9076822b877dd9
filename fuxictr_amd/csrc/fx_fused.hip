// fx_fused.hip — the sparse front end and back end of the training step as a handful of launches
// (BASELINE.json north_star: "the FeatureEmbedding multi-field sparse lookup ... fused straight into
// the feature-interaction layers").  Replaces, for one FeatureEmbeddingDict whose id columns own
// disjoint tables (every categorical schema of the BASELINE configs):
//
//   forward   fx_dedup_catchup   sort + (fused) begin-step  |  unique rows + exact-mode Adam catch-up of
//                                EVERY table group that shares the id plan (the D=16 tables and the
//                                D=1 tables of LogisticRegression) — 2 launches (were 6)
//             fx_emb_fm_fwd      gather + numeric expansion + first-order term + FM second-order term,
//                                one wave per sample, the record is written once — 1 launch (were 3)
//   backward  fx_emb_fm_bwd      FM backward folded into the run-reduce of the record's gradient, the
//                                D=1 rows reduced in the same pass, ||G||^2 partials fused, work
//                                balanced over the sorted lookups (no short/long split); numeric
//                                weights / LR bias in a second launch — 2 launches (were 13)
//   update    fx_sparse_adam_multi   the row update of all those table groups — 1 launch (were 2)
//   inputs    fx_pack_columns_multi  ids / numerics / label casts of a batch — 1 launch (were 3)
//
// Reference lines replaced (paths relative to the reference checkout):
//   fuxictr/pytorch/layers/embeddings/feature_embedding.py:261-297, :230-259 (lookup loop, stack)
//   fuxictr/pytorch/layers/blocks/logistic_regression.py:46-59 (second D=1 embedding pass, sum, bias)
//   fuxictr/pytorch/layers/interactions/inner_product.py:55-62 (product_sum) and their autograd,
//   aten::embedding_dense_backward at rank_model.py:320, the table part of clip_grad_norm_
//   (rank_model.py:321) and of torch.optim.Adam.step (rank_model.py:322).
#include "fx_common.h"


// ---------------------------------------------------------------------------------------------
// shared device pieces
// ---------------------------------------------------------------------------------------------

struct FxTableDev {
    void* table;           // fp32, or bf16 when `bf16` is set (moments / gradients are always fp32)
    float* m;
    float* v;
    int32_t* last_step;
    const float* G;        // update kernels only
    int32_t D, vec, lanes_log2, bf16;
    int64_t tld, mld, vld, lld;   // row strides of table / m / v (elements) and of last_step (ints): D, D, D, 1 for
                                  // packed arrays; all = W when the four point into one row record (round 6)
};

#define FX_MAX_TABLES 4

// zero-gradient Adam replay of one row (see k_adam_catchup), in two halves so that a lane group can
// issue the loads of EVERY table group before it waits for any of them: the rows live in multi-GB
// tables, every access is a TLB miss + an HBM access (~8 us per dependent round trip measured: the
// chain last_step -> m,v -> p of k_adam_catchup costs 25 us for 25 K rows), so the state of a row —
// last_step, m, v AND p, of the D-float table and of the D=1 table — is requested in one go.
template <int VEC>
struct FxRowRegs {
    float p[VEC], m[VEC], v[VEC];
    int last;
    bool on;       // this lane holds elements of the row
    bool act;      // this lane takes part at all (sub < lanes of the table)
};

template <int VEC, bool WANT_LAST = true>
__device__ __forceinline__ void fx_row_load(const FxTableDev& t, int64_t row, int sub,
                                            FxRowRegs<VEC>& r) {
    const int lanes = 1 << t.lanes_log2;
    r.act = sub < lanes;
    const int d0 = sub * VEC;
    r.on = r.act && d0 < t.D;
    r.last = 0;
#pragma unroll
    for (int k = 0; k < VEC; ++k) r.p[k] = r.m[k] = r.v[k] = 0.f;
    if (WANT_LAST && r.act) r.last = t.last_step[row * t.lld];
    if (r.on) {
        fx_load<VEC>(t.m + row * t.mld + d0, r.m);
        fx_load<VEC>(t.v + row * t.vld + d0, r.v);
        fx_tab_load<VEC>(t.table, t.bf16, row * t.tld + d0, r.p);
    }
}

template <int VEC>
__device__ __forceinline__ void fx_catchup_finish(const FxTableDev& t, int64_t row, int sub,
                                                  FxRowRegs<VEC>& r, const fx_scalars& sc, int upto,
                                                  const FxLogs& lg, const FxSeries& ser) {
    if (!r.act) return;
    const int last = r.last;
    const int k_steps = upto - last;
    if (k_steps <= 0) return;
    if (r.on) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < VEC; ++k) any = any || (r.m[k] != 0.f) || (r.v[k] != 0.f);
        if (any) {
            fx_adam_replay<VEC>(r.p, r.m, r.v, last, k_steps, sc, lg, ser);
            fx_tab_store<VEC>(t.table, t.bf16, row * t.tld + sub * VEC, r.p);
            fx_store<VEC>(t.m + row * t.mld + sub * VEC, r.m);
            fx_store<VEC>(t.v + row * t.vld + sub * VEC, r.v);
        }
    }
    if (sub == 0) t.last_step[row * t.lld] = upto;
}

template <int VEC>
__device__ __forceinline__ void fx_catchup_row(const FxTableDev& t, int64_t row, int sub,
                                               const fx_scalars& sc, int upto, const FxLogs& lg,
                                               const FxSeries& ser) {
    FxRowRegs<VEC> r;
    fx_row_load<VEC>(t, row, sub, r);
    fx_catchup_finish<VEC>(t, row, sub, r, sc, upto, lg, ser);
}

// ---------------------------------------------------------------------------------------------
// The DeepFM / xDeepFM shape of the catch-up — a 16-float row (4 lanes x 4 floats) and the D = 1 row of
// LogisticRegression under the same id — as ONE replay by the row's quad of lanes (round 5).
//
// Step i after `last` moves an element by  u_i = lr/(1-b1^(t+i)) . m b1^i / (sqrt(v) b2^(i/2) / sqrt(1-b2^(t+i)) + eps)
// (fx_adam_replay).  Factored:  u_i = (lr m / sqrt(v)) . w_i / (g_i + eps / sqrt(v))  with
//     w_i = b1^i / (1 - b1^(t+i)),   g_i = b2^(i/2) / sqrt(1 - b2^(t+i))
// the same for every element of the row: an element costs  acc += w_i . rcp(g_i + c)  per step (3 instructions,
// fx_adam_replay: 6), the sum is applied to p once.  (w_i, g_i) cost 10 instructions with two
// transcendentals: lane s of the quad computes them for step 4q + s + 1 of round q and the quad reads each
// other's pair through DPP quad broadcasts — 2.5 + 2 instructions a step instead of 10 on every lane.  The D = 1
// row rides as a fifth element (lane 0; zeros elsewhere) instead of a second pass with a quarter of the
// lanes.  19.75 instructions per step where the two passes of fx_adam_replay issued ~49 (k_catchup_rows
// is VALU-bound: a wave runs as long as its coldest row).
// The terms shrink by >= 5 % a step (b1 / sqrt(b2) over the ratio of the bias corrections), so once a
// step's terms are below 2^-29 of every sum of the quad the rest cannot change them in fp32: the quad is
// done; the wave leaves when all its quads are (no lane leaves the loop alone: there is no divergence).
// Against the reference's step-by-step `p -= u_i` this sums the same terms in the same order in fp32 and
// rounds p once instead of k times; what it drops is below 2^-29 of the move (tests: exact mode == dense
// torch Adam stepped k times).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fx_quad_bcast(float x, int u) {      // lane u of this lane's quad
    const int v = __float_as_int(x);
    int r;
    switch (u) {
        case 0: r = __builtin_amdgcn_mov_dpp(v, 0x00, 0xF, 0xF, true); break;      // quad_perm [0,0,0,0]
        case 1: r = __builtin_amdgcn_mov_dpp(v, 0x55, 0xF, 0xF, true); break;
        case 2: r = __builtin_amdgcn_mov_dpp(v, 0xAA, 0xF, 0xF, true); break;
        default: r = __builtin_amdgcn_mov_dpp(v, 0xFF, 0xF, 0xF, true); break;
    }
    return __int_as_float(r);
}

__device__ __forceinline__ bool fx_quad_any(bool x) {
    const unsigned long long b = __ballot(x);
    const int q4 = (threadIdx.x & 63) & ~3;
    return ((b >> q4) & 0xFull) != 0ull;
}

// LR = false: the D = 16 table alone (DCNv2, DLRM, ...: models without a first-order term).
// r0 / r1 leave with the row as it stands after the catch-up (the owner fetch of the row-sharded path sends it
// from there: one code path, one rounding, for 1 rank and for N).
template <bool LR>
__device__ __forceinline__ void fx_catchup_quad(const FxTableDev& t0, const FxTableDev& t1, int64_t row, int sub,
                                                const fx_scalars& sc, int upto, const FxLogs& lg,
                                                const FxSeries& ser, FxRowRegs<4>& r0, FxRowRegs<1>& r1) {
    fx_row_load<4, false>(t0, row, sub, r0);
    if constexpr (LR) fx_row_load<1, false>(t1, row, sub, r1);
    else { r1.p[0] = r1.m[0] = r1.v[0] = 0.f; r1.on = r1.act = false; r1.last = 0; }
    const int last = t0.last_step[row * t0.lld];      // (same address in the four lanes: one access)
    const int last1 = LR ? t1.last_step[row * t1.lld] : last;
    const int k0 = upto - last, k1 = upto - last1;
    if (last1 != last) {
        // the two tables were not touched together (cannot happen under one id plan): the plain replays
        r0.last = last; r1.last = last1;
        fx_catchup_finish<4>(t0, row, sub, r0, sc, upto, lg, ser);
        fx_catchup_finish<1>(t1, row, sub, r1, sc, upto, lg, ser);
        return;
    }
    if (k0 <= 0) return;                               // (the whole quad: `last` is the row's)
    (void)k1;
    // this lane's five elements: 4 of the D-float row + the D = 1 row (lane 0)
    float pe[5], me[5], ve[5];
#pragma unroll
    for (int e = 0; e < 4; ++e) { pe[e] = r0.p[e]; me[e] = r0.m[e]; ve[e] = r0.v[e]; }
    pe[4] = r1.p[0]; me[4] = r1.m[0]; ve[4] = r1.v[0];
    bool any_state = false, moving = false;
#pragma unroll
    for (int e = 0; e < 5; ++e) {
        any_state |= (me[e] != 0.f) || (ve[e] != 0.f);
        moving |= (me[e] != 0.f);
    }
    const int kk = k0 < FX_REPLAY_MAX ? k0 : FX_REPLAY_MAX;
    if (ser.tab != nullptr && k0 > FX_SERIES_KDIR) {
        // (round 6) the sum of the missed steps from the series table: no step loop (fx_common.h)
        if (moving) fx_series_move<5>(pe, me, ve, last, k0, sc, ser, lg);
    } else if (fx_quad_any(moving)) {
        float c[5], sc_e[5], acc[5];
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            const float r = sqrtf(ve[e]);
            const bool live = (me[e] != 0.f) && (r > 0.f);
            const float inv = live ? 1.f / r : 0.f;
            c[e] = live ? sc.eps * inv : 1.f;          // (a dead element: acc grows harmlessly, scale 0)
            sc_e[e] = sc.lr * me[e] * inv;
            acc[e] = 0.f;
        }
        // lane `sub` owns the steps 4 q + sub + 1
        const float b1 = sc.beta1, b2 = sc.beta2, sb2 = sqrtf(sc.beta2);
        const float b1_2 = b1 * b1, b2_2 = b2 * b2, sb2_2 = sb2 * sb2;
        const float b1_4 = b1_2 * b1_2, b2_4 = b2_2 * b2_2, sb2_4 = sb2_2 * sb2_2;
        float bi = sub == 0 ? b1 : sub == 1 ? b1_2 : sub == 2 ? b1_2 * b1 : b1_4;          // b1^(sub+1)
        float sb = sub == 0 ? sb2 : sub == 1 ? sb2_2 : sub == 2 ? sb2_2 * sb2 : sb2_4;
        // the bias corrections as d = 1 - b^(t+i), advanced by d' = (1 - b^4) + b^4 d (round 6: 1 - b2^t is
        // 0.001 t early in a run; b2^t rounded to fp32 first left it with a relative error of 3e-5 / t)
        // (lc*: torch's python-side doubles — fx_beta_f64)
        float d1 = (float)(1.0 - exp2(lg.lc1 * (double)(last + sub + 1)));
        float d2 = (float)(1.0 - exp2(lg.lc2 * (double)(last + sub + 1)));
        const float e1_4 = (float)(1.0 - exp2(4.0 * lg.lc1)), e2_4 = (float)(1.0 - exp2(4.0 * lg.lc2));
        const int nr = (kk + 3) >> 2;
        for (int q = 0; q < nr; ++q) {
            const int i = 4 * q + sub + 1;
            float w = bi * __builtin_amdgcn_rcpf(d1);
            const float g = sb * __builtin_amdgcn_rsqf(d2);
            w = i <= kk ? w : 0.f;
            bi *= b1_4; sb *= sb2_4;
            d1 = fmaf(b1_4, d1, e1_4); d2 = fmaf(b2_4, d2, e2_4);
            float term[5];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float wu = fx_quad_bcast(w, u), gu = fx_quad_bcast(g, u);
#pragma unroll
                for (int e = 0; e < 5; ++e) {
                    term[e] = wu * __builtin_amdgcn_rcpf(gu + c[e]);
                    acc[e] += term[e];
                }
            }
            // (term[] = the round's last step.)  Done when it can no longer change any live sum of the quad.
            bool small = true;
#pragma unroll
            for (int e = 0; e < 5; ++e) small &= (sc_e[e] == 0.f) || (term[e] <= acc[e] * 1.862645e-9f);   // 2^-29
            // (quads that have counted all their steps have left the loop and do not vote; the ones still
            // here leave together)
            if (__all(!fx_quad_any(!small))) break;
        }
#pragma unroll
        for (int e = 0; e < 5; ++e) pe[e] = fmaf(-sc_e[e], acc[e], pe[e]);
    }
    // the decay of the moments over ALL missed steps, in closed form
    if (any_state) {
        const float f1 = (float)exp2(lg.lb1 * (double)k0), f2 = (float)exp2(lg.lb2 * (double)k0);
#pragma unroll
        for (int e = 0; e < 5; ++e) { me[e] *= f1; ve[e] *= f2; }
    }
    if (r0.on) {
        bool any = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) any |= (r0.m[e] != 0.f) || (r0.v[e] != 0.f);
        if (any) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { r0.p[e] = pe[e]; r0.m[e] = me[e]; r0.v[e] = ve[e]; }
            fx_tab_store<4>(t0.table, t0.bf16, row * t0.tld + sub * 4, r0.p);
            fx_store<4>(t0.m + row * t0.mld + sub * 4, r0.m);
            fx_store<4>(t0.v + row * t0.vld + sub * 4, r0.v);
        }
    }
    if (r1.on && ((r1.m[0] != 0.f) || (r1.v[0] != 0.f))) {
        r1.p[0] = pe[4]; r1.m[0] = me[4]; r1.v[0] = ve[4];
        fx_tab_store<1>(t1.table, t1.bf16, row * t1.tld, r1.p);
        fx_store<1>(t1.m + row * t1.mld, r1.m);
        fx_store<1>(t1.v + row * t1.vld, r1.v);
    }
    if (sub == 0) {
        t0.last_step[row * t0.lld] = upto;
        if constexpr (LR) t1.last_step[row * t1.lld] = upto;
    }
}

static const bool fx_catchup_quad_on = []() {     // FX_CATCHUP_QUAD=0: the two plain replays (A/B runs)
    const char* e = getenv("FX_CATCHUP_QUAD");
    return !(e && atoi(e) == 0);
}();

// one unique row of a de-dup result, in every table group that shares the id plan
__device__ __forceinline__ void fx_catchup_tables(const FxTableDev* t, int n_tables, int64_t row,
                                                  int sub, const fx_scalars& sc, int upto, const FxLogs& lg,
                                                  const FxSeries& ser) {
    if (n_tables == 2 && t[0].vec == 4 && t[1].vec == 1) {
        // the D-float tables + the D=1 tables of LogisticRegression: all eight loads in flight
        FxRowRegs<4> r0;
        FxRowRegs<1> r1;
        fx_row_load<4>(t[0], row, sub, r0);
        fx_row_load<1>(t[1], row, sub, r1);
        fx_catchup_finish<4>(t[0], row, sub, r0, sc, upto, lg, ser);
        fx_catchup_finish<1>(t[1], row, sub, r1, sc, upto, lg, ser);
        return;
    }
    for (int i = 0; i < n_tables; ++i) {
        const FxTableDev& tb = t[i];
        if (tb.vec == 4) fx_catchup_row<4>(tb, row, sub, sc, upto, lg, ser);
        else if (tb.vec == 2) fx_catchup_row<2>(tb, row, sub, sc, upto, lg, ser);
        else fx_catchup_row<1>(tb, row, sub, sc, upto, lg, ser);
    }
}

// ---------------------------------------------------------------------------------------------
// fx_dedup_catchup, launch 1: one workgroup sorts one id column in LDS — a hand-written stable LSD
// radix sort, 8 bits a pass over only the bits the column's vocabulary uses (a 10 M-row table: 3
// passes; the many Criteo columns with < 256 values: 1), 16 waves x IPT rounds of 64 keys:
//   rank    per round, the lanes that hold the same digit find each other with 8 ballots (one per
//           digit bit); a key's rank inside its wave's chunk = the wave's running count of that digit
//           (a private 256-entry histogram row in LDS, read by every lane of the match group, bumped
//           by its lowest lane) + the number of lower lanes of the group — chunk order is index order,
//           so the sort is stable and the order of a row's lookups (hence its gradient sum) is fixed;
//   offsets 256 threads turn the 16 histogram rows into exclusive per-wave offsets and scan the digit
//           totals (wave shuffles);
//   scatter key and position move to the other LDS buffer.
// Then head flags, a block scan of them, and the sorted column leaves for global memory with its
// running unique count.  Block 0 also opens the optimizer step (fx_opt_begin_step fused).
// (Round 2 used rocprim::block_radix_sort here — 4 bits a pass, 23 us for the 26 Criteo columns.)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t fx_wave_incl_scan_u32(uint32_t x, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(x, off, 64);
        if (lane >= off) x += y;
    }
    return x;
}

template <int IPT>
__global__ __launch_bounds__(1024) void k_sort_columns3(const int32_t* ids, int64_t ids_ld, int64_t B,
                                                        const int64_t* col_row_base,
                                                        const int32_t* col_vocab,
                                                        const int32_t* col_pad, int C,
                                                        uint32_t* sorted_key, uint32_t* sorted_pos,
                                                        uint32_t* col_scan, uint32_t* col_cnt,
                                                        fx_scalars* begin_scal) {
    constexpr int N = IPT * 1024;
    __shared__ uint32_t kbuf[2][N];
    __shared__ uint32_t pbuf[2][N];
    __shared__ uint32_t hist[16][256];
    __shared__ uint32_t dbase[256];
    __shared__ uint32_t wtot[16];
    const int c = blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (begin_scal != nullptr && c == 0 && threadIdx.x == 0) fx_begin_step_dev(begin_scal);
    const int32_t V = col_vocab[c], pad = col_pad[c];
    int bits = 1;
    while ((1u << bits) < (uint32_t)V && bits < 31) ++bits;
    const uint32_t fill = (1u << bits) - 1u;   // >= every real id; ties keep real items first
    // load: item i = position b of the column (coalescing does not matter: one 4-byte id per row)
#pragma unroll
    for (int r = 0; r < IPT; ++r) {
        const int i = r * 1024 + threadIdx.x;
        uint32_t k = fill, v = 0xFFFFFFFFu;
        if (i < B) {
            const int32_t id = ids[(int64_t)i * ids_ld + c];
            const bool in_range = id >= 0 && id < V;
            k = in_range ? (uint32_t)id : 0u;
            if (in_range && id != pad) v = (uint32_t)((int64_t)i * C + c);
        }
        kbuf[0][i] = k;
        pbuf[0][i] = v;
    }
    __syncthreads();
    int src = 0;
    for (int shift = 0; shift < bits; shift += 8) {
        // -- rank inside the wave's chunk
#pragma unroll
        for (int j = 0; j < 4; ++j) hist[w][lane + 64 * j] = 0u;
        uint32_t kk[IPT], pp[IPT], rk[IPT];
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const int i = (w * IPT + r) * 64 + lane;
            kk[r] = kbuf[src][i];
            pp[r] = pbuf[src][i];
            const uint32_t d = (kk[r] >> shift) & 255u;
            uint64_t m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bal = __ballot(bit);
                m &= bit ? bal : ~bal;
            }
            const uint32_t below = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            const uint32_t base = hist[w][d];           // every lane of the group reads the same word
            __builtin_amdgcn_wave_barrier();
            if (below == 0u) hist[w][d] = base + (uint32_t)__popcll(m);
            __builtin_amdgcn_wave_barrier();
            rk[r] = base + below;
        }
        __syncthreads();
        // -- per-wave exclusive offsets of every digit, digit totals, their exclusive scan
        if (threadIdx.x < 256) {
            const int d = threadIdx.x;
            uint32_t run = 0;
#pragma unroll
            for (int ww = 0; ww < 16; ++ww) {
                const uint32_t t = hist[ww][d];
                hist[ww][d] = run;
                run += t;
            }
            const uint32_t inc = fx_wave_incl_scan_u32(run, lane);
            if (lane == 63) wtot[w] = inc;
            dbase[d] = inc - run;                        // exclusive inside the wave
        }
        __syncthreads();
        if (threadIdx.x < 256) {
            uint32_t add = 0;
            for (int ww = 0; ww < w; ++ww) add += wtot[ww];
            dbase[threadIdx.x] += add;
        }
        __syncthreads();
        // -- scatter
#pragma unroll
        for (int r = 0; r < IPT; ++r) {
            const uint32_t d = (kk[r] >> shift) & 255u;
            const uint32_t dst = dbase[d] + hist[w][d] + rk[r];
            kbuf[src ^ 1][dst] = kk[r];
            pbuf[src ^ 1][dst] = pp[r];
        }
        __syncthreads();
        src ^= 1;
    }
    // head flags over the sorted column (thread t owns the IPT consecutive items t*IPT ...), block scan
    uint32_t k[IPT], flag[IPT], h = 0;
#pragma unroll
    for (int j = 0; j < IPT; ++j) k[j] = kbuf[src][threadIdx.x * IPT + j];
    const uint32_t prev0 = threadIdx.x > 0 ? kbuf[src][threadIdx.x * IPT - 1] : 0u;
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = (int64_t)threadIdx.x * IPT + j;
        const uint32_t prev = j > 0 ? k[j - 1] : prev0;
        flag[j] = (i < B && (i == 0 || k[j] != prev)) ? 1u : 0u;
        h += flag[j];
    }
    const uint32_t inc = fx_wave_incl_scan_u32(h, lane);
    if (lane == 63) wtot[w] = inc;
    __syncthreads();
    uint32_t before = inc - h;
    for (int ww = 0; ww < w; ++ww) before += wtot[ww];
    if (threadIdx.x == 1023) col_cnt[c] = before + h;
    const uint32_t base = (uint32_t)col_row_base[c];
#pragma unroll
    for (int j = 0; j < IPT; ++j) {
        const int64_t i = (int64_t)threadIdx.x * IPT + j;
        before += flag[j];
        if (i < B) {
            sorted_key[(int64_t)c * B + i] = base + k[j];
            sorted_pos[(int64_t)c * B + i] = pbuf[src][i];
            col_scan[(int64_t)c * B + i] = before;
        }
    }
}

// launch 2: one lane group per sorted lookup.  The group of a run's FIRST lookup owns the unique
// row: it writes uniq_row / seg_start and replays the row's missed zero-gradient Adam steps in
// every table group that shares the id plan, so the gather that follows reads current rows.
struct FinishArgs {
    const uint32_t* key;
    const uint32_t* col_scan;
    const uint32_t* col_cnt;
    uint32_t* uniq_row;
    uint32_t* seg_start;
    int32_t* n_unique;
    uint32_t* sorted_uid;
    FxTableDev t[FX_MAX_TABLES];
    const fx_scalars* scal;
    int64_t B;
    int32_t C, n_tables, group_log2, upto_offset;
};

__global__ __launch_bounds__(256) void k_finish_catchup(FinishArgs a) {
    __shared__ uint32_t off[257];
    if (threadIdx.x == 0) {
        uint32_t acc = 0;
        for (int c = 0; c < a.C; ++c) {
            off[c] = acc;
            acc += a.col_cnt[c];
        }
        off[a.C] = acc;
    }
    __syncthreads();
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t ipb = 256 >> a.group_log2;
    const int64_t n = a.B * a.C;
    fx_scalars sc;
    int upto = 0;
    FxLogs lg{0.0, 0.0, 0.0, 0.0};
    FxSeries ser{nullptr, 0};
    if (a.n_tables > 0) {
        sc = *a.scal;
        upto = sc.step + a.upto_offset;
        lg = fx_logs_of(sc);
        ser = fx_series_of(a.scal, sc);
    }
    for (int64_t i = (int64_t)blockIdx.x * ipb + (threadIdx.x >> a.group_log2); i < n;
         i += (int64_t)gridDim.x * ipb) {
        const int c = (int)(i / a.B);
        const uint32_t k = a.key[i];
        const uint32_t u = off[c] + a.col_scan[i];
        const bool head = (i == 0) || (a.key[i - 1] != k);
        if (sub == 0) {
            if (a.sorted_uid) a.sorted_uid[i] = u - 1;
            if (head) {
                a.uniq_row[u - 1] = k;
                a.seg_start[u - 1] = (uint32_t)i;
            }
            if (i == n - 1) {
                a.seg_start[u] = (uint32_t)(i + 1);
                *a.n_unique = (int32_t)u;
            }
        }
        if (!head) continue;
        fx_catchup_tables(a.t, a.n_tables, (int64_t)k, sub, sc, upto, lg, ser);
    }
}

static int fx_fill_tables(const fx_row_state* tables_host, int32_t n_tables, FxTableDev* out,
                          int* group_log2, const char* who, bool need_state) {
    int gl = 0;
    for (int t = 0; t < n_tables; ++t) {
        const fx_row_state& h = tables_host[t];
        if (h.D < 1 || h.D > 256) {
            fx_set_error("%s: table %d has D=%d outside [1,256]", who, t, h.D);
            return FX_ERR_INVALID;
        }
        if (!h.table || (need_state && (!h.m || !h.v || !h.last_step))) {
            fx_set_error("%s: table %d has a null pointer", who, t);
            return FX_ERR_INVALID;
        }
        const FxRowGeom g = fx_row_geom(h.D);
        int ll = 0;
        while ((1 << ll) < g.lanes) ++ll;
        out[t].table = h.table;
        out[t].m = h.m;
        out[t].v = h.v;
        out[t].last_step = h.last_step;
        out[t].G = h.G;
        out[t].D = h.D;
        out[t].tld = h.table_ld > 0 ? h.table_ld : h.D;
        out[t].mld = h.m_ld > 0 ? h.m_ld : h.D;
        out[t].vld = h.v_ld > 0 ? h.v_ld : h.D;
        out[t].lld = h.last_ld > 0 ? h.last_ld : 1;
        if (out[t].tld < h.D || out[t].mld < h.D || out[t].vld < h.D) {
            fx_set_error("%s: table %d has a row stride below D", who, t);
            return FX_ERR_INVALID;
        }
        out[t].bf16 = h.table_dtype == FX_BF16 ? 1 : 0;
        if (h.table_dtype != FX_F32 && h.table_dtype != FX_BF16) {
            fx_set_error("%s: table %d has table_dtype %d (FX_F32 or FX_BF16)", who, t, h.table_dtype);
            return FX_ERR_INVALID;
        }
        out[t].vec = g.vec;
        out[t].lanes_log2 = ll;
        if (ll > gl) gl = ll;
    }
    if (gl > 6) {
        fx_set_error("%s: rows of more than 64 lanes are not supported here", who);
        return FX_ERR_UNSUPPORTED;
    }
    *group_log2 = gl;
    return FX_OK;
}

static int fx_launch_catchup_rows(const FxTableDev* t, int n_tables, int gl, const uint32_t* uniq_row,
                                  const int32_t* n_unique, int64_t n_max, int32_t upto_offset,
                                  const fx_scalars* scal, hipStream_t s);

static void fx_launch_sort_columns(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                                   const int64_t* col_row_base, const int32_t* col_vocab,
                                   const int32_t* col_pad, uint32_t* sorted_key, uint32_t* sorted_pos,
                                   uint32_t* col_scan, uint32_t* col_cnt, fx_scalars* begin_scal,
                                   hipStream_t s) {
#define FX_SORT3(IPT)                                                                             \
    hipLaunchKernelGGL(k_sort_columns3<IPT>, dim3(C), dim3(1024), 0, s, ids, ids_ld, B,           \
                       col_row_base, col_vocab, col_pad, (int)C, sorted_key, sorted_pos, col_scan, \
                       col_cnt, begin_scal)
    if (B <= 1024) FX_SORT3(1);
    else if (B <= 2048) FX_SORT3(2);
    else if (B <= 4096) FX_SORT3(4);
    else FX_SORT3(8);
#undef FX_SORT3
}

// the column fast path without tables, for fx_dedup (declared in fx_common.h)
int fx_dedup_columns_launch(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                            const int64_t* col_row_base, const int32_t* col_vocab,
                            const int32_t* col_pad, uint32_t* col_cnt, uint32_t* col_scan,
                            uint32_t* sorted_key, uint32_t* sorted_pos, uint32_t* uniq_row,
                            uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                            fx_scalars* begin_scal, hipStream_t s) {
    fx_launch_sort_columns(ids, ids_ld, B, C, col_row_base, col_vocab, col_pad, sorted_key, sorted_pos,
                           col_scan, col_cnt, begin_scal, s);
    FX_CHECK_LAUNCH();
    FinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    fa.key = sorted_key;
    fa.col_scan = col_scan;
    fa.col_cnt = col_cnt;
    fa.uniq_row = uniq_row;
    fa.seg_start = seg_start;
    fa.n_unique = n_unique;
    fa.sorted_uid = sorted_uid;
    fa.B = B;
    fa.C = C;
    const int64_t n = B * (int64_t)C;
    int64_t blocks = fx_ceil_div(n, 256);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_finish_catchup, dim3((unsigned)blocks), dim3(256), 0, s, fa);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_dedup_catchup(const int32_t* ids, int64_t ids_ld, int64_t B, int32_t C,
                                const int64_t* col_row_base, const int32_t* col_vocab,
                                const int32_t* col_pad, void* workspace, size_t workspace_bytes,
                                uint32_t* sorted_key, uint32_t* sorted_pos, uint32_t* uniq_row,
                                uint32_t* seg_start, int32_t* n_unique, uint32_t* sorted_uid,
                                fx_scalars* begin_scal, const fx_row_state* tables_host,
                                int32_t n_tables, int32_t upto_offset, const fx_scalars* scal,
                                fx_stream_t stream) {
    FX_CHECK_ARG(B >= 1 && B <= 8192 && C >= 1 && C <= 256,
                 "fx_dedup_catchup: B=%lld (1..8192) / C=%d (1..256) outside the column fast path",
                 (long long)B, C);
    FX_CHECK_ARG(n_tables >= 0 && n_tables <= FX_MAX_TABLES, "fx_dedup_catchup: n_tables=%d > %d",
                 n_tables, FX_MAX_TABLES);
    FX_CHECK_ARG(ids && col_row_base && col_vocab && col_pad && workspace && sorted_key &&
                     sorted_pos && uniq_row && seg_start && n_unique,
                 "fx_dedup_catchup: null pointer");
    FX_CHECK_ARG(n_tables == 0 || (tables_host && scal), "fx_dedup_catchup: tables without scal");
    const int64_t n = B * (int64_t)C;
    const size_t arr = ((size_t)n * sizeof(uint32_t) + 255) / 256 * 256;
    FX_CHECK_ARG(workspace_bytes >= 2 * arr, "fx_dedup_catchup: workspace too small (%zu < %zu)",
                 workspace_bytes, 2 * arr);
    char* w = reinterpret_cast<char*>(workspace);
    uint32_t* col_cnt = reinterpret_cast<uint32_t*>(w);
    uint32_t* col_scan = reinterpret_cast<uint32_t*>(w + arr);
    FinishArgs fa;
    memset(&fa, 0, sizeof(fa));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, fa.t, &gl, "fx_dedup_catchup", true);
    if (st != FX_OK) return st;
    hipStream_t s = fx_hip_stream(stream);
    fx_launch_sort_columns(ids, ids_ld, B, C, col_row_base, col_vocab, col_pad, sorted_key, sorted_pos,
                           col_scan, col_cnt, begin_scal, s);
    FX_CHECK_LAUNCH();
    fa.key = sorted_key;
    fa.col_scan = col_scan;
    fa.col_cnt = col_cnt;
    fa.uniq_row = uniq_row;
    fa.seg_start = seg_start;
    fa.n_unique = n_unique;
    fa.sorted_uid = sorted_uid;
    fa.scal = scal;
    fa.B = B;
    fa.C = C;
    // The catch-up of the unique rows runs as its OWN dense launch over uniq_row (FX_SPLIT_CATCHUP=0: inside
    // the scan / scatter launch, as in round 2): there a lane group exists per LOOKUP and only the 24 % that
    // head a run have a row to replay — 99 % of the waves walked the row path with a quarter of their
    // lanes; over the compacted rows every lane works (profiles/r03_sparse_ab.txt).
    static const bool split = []() {
        const char* e = getenv("FX_SPLIT_CATCHUP");
        return !(e && atoi(e) == 0);
    }();
    const bool two = split && n_tables > 0;
    fa.n_tables = two ? 0 : n_tables;
    fa.group_log2 = two ? 0 : gl;
    fa.upto_offset = upto_offset;
    int64_t blocks = fx_ceil_div(n, 256 >> fa.group_log2);
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(k_finish_catchup, dim3((unsigned)blocks), dim3(256), 0, s, fa);
    FX_CHECK_LAUNCH();
    if (two) return fx_launch_catchup_rows(fa.t, n_tables, gl, uniq_row, n_unique, n, upto_offset, scal, s);
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_fm_fwd: one wave per sample.  The wave's 64 lanes are 64/lanes lane groups; group g walks
// the record's items g, g+G, ... (an id lookup or a numeric expansion), writes the item's row into
// its slot of the record and keeps the per-dimension field sums and the sum of squares in
// registers; the first-order term is summed by lane l over id column l / numeric column l.  Two
// xor butterflies finish the sample: nothing but the record, the field sums S (for the backward) and
// three scalars per sample is written.
// ---------------------------------------------------------------------------------------------
struct EmbFmArgs {
    const void* table;     // fp32, or bf16 when `bf16` is set
    const int32_t* ids;
    int64_t ids_ld;
    const int64_t* col_row_base;
    const int32_t* col_vocab;
    const int64_t* col_out_off;
    const float* dense;
    int64_t dense_ld;
    const float* num_w;
    const int64_t* num_out_off;
    float* out;
    int64_t out_ld;
    int64_t B;
    const float* table1;
    const float* num_w1;
    const float* bias1;
    float* lr_out;
    float* fm_out;
    float* fm_lr_out;
    float* S;
    fx_scalars* scal;
    int32_t D, C, Fd, lanes_log2, bf16;
    int64_t table_ld, table1_ld;     // row strides in elements (D and 1 for packed tables; the block of
                                     // rows a row-sharded exchange delivered is read in place)
    int64_t zero_off[2];             // reserved slots of the record (a DIN hole, DLRM's tail): zero_n floats
    int32_t zero_n[2];               // at zero_off of every sample's row are cleared (their producer writes
                                     // them later in the step; nobody may ever read uninitialised memory)
};

__device__ __forceinline__ void fx_zero_reserved(const EmbFmArgs& a, int64_t b, int lane) {
#pragma unroll
    for (int r = 0; r < 2; ++r)
        for (int i = lane; i < a.zero_n[r]; i += 64) a.out[b * a.out_ld + a.zero_off[r] + i] = 0.f;
}

template <int VEC>
__global__ __launch_bounds__(256) void k_emb_fm_fwd(EmbFmArgs a) {
    const int lane = threadIdx.x & 63;
    const int lanes = 1 << a.lanes_log2;
    const int sub = lane & (lanes - 1);
    const int grp = lane >> a.lanes_log2;
    const int ngrp = 64 >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int R = a.C + a.Fd;
    const bool want_fm = a.fm_out != nullptr || a.fm_lr_out != nullptr || a.S != nullptr;
    for (int64_t b = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); b < a.B;
         b += (int64_t)gridDim.x * 4) {                                  // wave-uniform
        float s[VEC], q = 0.f;
#pragma unroll
        for (int k = 0; k < VEC; ++k) s[k] = 0.f;
        for (int r = grp; r < R; r += ngrp) {
            float val[VEC];
#pragma unroll
            for (int k = 0; k < VEC; ++k) val[k] = 0.f;
            int64_t off;
            if (r < a.C) {
                const int32_t id = a.ids[b * a.ids_ld + r];
                off = a.col_out_off[r];
                if (id >= 0 && id < a.col_vocab[r]) {
                    if (lane_on)
                        fx_tab_load<VEC>(a.table, a.bf16, (a.col_row_base[r] + id) * a.table_ld + d0, val);
                } else if (sub == 0) {
                    atomicOr(&a.scal->err_flag, FX_FLAG_BAD_ID);
                }
            } else {
                const int j = r - a.C;
                off = a.num_out_off[j];
                if (lane_on) {
                    const float x = a.dense[b * a.dense_ld + j];
                    float w[VEC];
                    fx_load<VEC>(a.num_w + (int64_t)j * a.D + d0, w);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) val[k] = x * w[k];
                }
            }
            if (lane_on) {
                fx_store<VEC>(a.out + b * a.out_ld + off + d0, val);
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    s[k] += val[k];
                    q = fmaf(val[k], val[k], q);
                }
            }
        }
        float lr = 0.f;
        if (a.lr_out != nullptr || a.fm_lr_out != nullptr) {
            for (int c = lane; c < a.C; c += 64) {
                const int32_t id = a.ids[b * a.ids_ld + c];
                if (id >= 0 && id < a.col_vocab[c]) lr += a.table1[(a.col_row_base[c] + id) * a.table1_ld];
            }
            for (int j = lane; j < a.Fd; j += 64)
                lr = fmaf(a.dense[b * a.dense_ld + j], a.num_w1[j], lr);
            lr = fx_wave_sum(lr);
            if (a.bias1) lr += a.bias1[0];
        }
        float fm = 0.f;
        if (want_fm) {
            // per-dimension field sums: combine the lane groups (fixed butterfly order)
            for (int o = lanes; o < 64; o <<= 1) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) s[k] += __shfl_xor(s[k], o, 64);
            }
            float t = -q;
            if (grp == 0 && lane_on) {
#pragma unroll
                for (int k = 0; k < VEC; ++k) t = fmaf(s[k], s[k], t);
            }
            fm = 0.5f * fx_wave_sum(t);
            if (a.S && grp == 0 && lane_on) fx_store<VEC>(a.S + b * a.D + d0, s);
        }
        if (lane == 0) {
            if (a.lr_out) a.lr_out[b] = lr;
            if (a.fm_out) a.fm_out[b] = fm;
            if (a.fm_lr_out) a.fm_lr_out[b] = fm + lr;
        }
        fx_zero_reserved(a, b, lane);
    }
}

// Round 4: the same work with every load of a sample in flight at once.  The first version walked
// id -> col_vocab -> col_row_base -> row -> store three times in sequence per lane group and started the
// 26 scattered 4-byte loads of the first-order term only after the record loop: 3 - 4 dependent HBM
// round trips per sample, 10.8 us at B = 4096 (0.18 of the HBM roofline).  Here
//   * the per-column constants (vocabulary, row base, record offset, the numeric weight rows) of the
//     NI items a lane group serves are loaded ONCE per wave, outside the sample loop;
//   * lane l loads id l / numeric l of the sample — one coalesced 104-byte + 52-byte request per
//     sample — and the lane groups fetch "their" ids by shuffle;
//   * all NI row loads of the record AND the lane's D = 1 row of the first-order term are issued
//     before the first use, the stores follow;
//   * SPW samples per wave and iteration (2 at large B) double the rows in flight.
// Shapes: C <= 64 id columns, Fd <= 64 numerics, C + Fd <= NI * (64 / lanes), NI <= 4 — every schema
// of the BASELINE configs; anything else keeps k_emb_fm_fwd.  Same sums in the same order: bit-identical
// outputs (tests/test_gpu_fused.py::test_emb_fm_fwd_equals_the_three_unfused_kernels and the strided /
// two-sample tests).
template <int VEC, int NI, int SPW>
__global__ __launch_bounds__(256) void k_emb_fm_fwd2(EmbFmArgs a) {
    const int lane = threadIdx.x & 63;
    const int lanes = 1 << a.lanes_log2;
    const int sub = lane & (lanes - 1);
    const int grp = lane >> a.lanes_log2;
    const int ngrp = 64 >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int R = a.C + a.Fd;
    const bool want_fm = a.fm_out != nullptr || a.fm_lr_out != nullptr || a.S != nullptr;
    const bool want_lr = a.lr_out != nullptr || a.fm_lr_out != nullptr;
    // ---- per-wave constants of this lane group's items
    int kind[NI];                 // 0: none, 1: id column, 2: numeric column
    int32_t vocab[NI];
    int64_t base[NI], off[NI];
    float wnum[NI][VEC];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = grp + j * ngrp;
        kind[j] = r < a.C ? 1 : (r < R ? 2 : 0);
        vocab[j] = 0;
        base[j] = off[j] = 0;
#pragma unroll
        for (int k = 0; k < VEC; ++k) wnum[j][k] = 0.f;
        if (kind[j] == 1) {
            vocab[j] = a.col_vocab[r];
            base[j] = a.col_row_base[r];
            off[j] = a.col_out_off[r];
        } else if (kind[j] == 2) {
            off[j] = a.num_out_off[r - a.C];
            if (lane_on) fx_load<VEC>(a.num_w + (int64_t)(r - a.C) * a.D + d0, wnum[j]);
        }
    }
    // ... and of the lane's own column (ids of the sample, first-order term)
    const bool has_id = lane < a.C, has_x = lane < a.Fd;
    int32_t my_vocab = 0;
    int64_t my_base = 0;
    float my_w1 = 0.f;
    if (has_id) {
        my_vocab = a.col_vocab[lane];
        my_base = a.col_row_base[lane];
    }
    if (want_lr && has_x) my_w1 = a.num_w1[lane];
    const float bias1 = (want_lr && a.bias1) ? a.bias1[0] : 0.f;
    const int64_t wstride = (int64_t)gridDim.x * 4 * SPW;
    for (int64_t b0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * SPW; b0 < a.B; b0 += wstride) {
        int32_t idl[SPW];
        float xl[SPW];
        bool live[SPW];
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const int64_t b = b0 + s;
            live[s] = b < a.B;                                            // wave-uniform
            idl[s] = -1;
            xl[s] = 0.f;
            if (live[s] && has_id) idl[s] = a.ids[b * a.ids_ld + lane];
            if (live[s] && has_x) xl[s] = a.dense[b * a.dense_ld + lane];
        }
        // ---- all row loads of the sample(s)
        float val[SPW][NI][VEC];
        float t1[SPW];
        bool bad = false;
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            const bool ok_l = has_id && idl[s] >= 0 && idl[s] < my_vocab;
            t1[s] = 0.f;
            if (live[s] && has_id && !ok_l) bad = true;
            if (want_lr && live[s] && ok_l) t1[s] = a.table1[(my_base + idl[s]) * a.table1_ld];
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int r = grp + j * ngrp;
#pragma unroll
                for (int k = 0; k < VEC; ++k) val[s][j][k] = 0.f;
                // (the shuffles are executed by every lane: the source lane index is r or r - C)
                const int32_t id = __shfl(idl[s], r < a.C ? r : 0, 64);
                const float x = __shfl(xl[s], r >= a.C && r < R ? r - a.C : 0, 64);
                if (!live[s]) continue;
                if (kind[j] == 1) {
                    if (id >= 0 && id < vocab[j] && lane_on)
                        fx_tab_load<VEC>(a.table, a.bf16, (base[j] + id) * a.table_ld + d0, val[s][j]);
                } else if (kind[j] == 2) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) val[s][j][k] = x * wnum[j][k];
                }
            }
        }
        if (bad) atomicOr(&a.scal->err_flag, FX_FLAG_BAD_ID);
        // ---- record stores, field sums, first-order term
#pragma unroll
        for (int s = 0; s < SPW; ++s) {
            if (!live[s]) continue;
            const int64_t b = b0 + s;
            float sm[VEC], q = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) sm[k] = 0.f;
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                if (kind[j] != 0 && lane_on) {
                    fx_store<VEC>(a.out + b * a.out_ld + off[j] + d0, val[s][j]);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) {
                        sm[k] += val[s][j][k];
                        q = fmaf(val[s][j][k], val[s][j][k], q);
                    }
                }
            }
            float lr = 0.f;
            if (want_lr) {
                // (the first version summed ids, then numerics, per lane: lane l holds column l of both)
                lr = t1[s];
                if (has_x) lr = fmaf(xl[s], my_w1, lr);
                lr = fx_wave_sum(lr) + bias1;
            }
            float fm = 0.f;
            if (want_fm) {
                for (int o = lanes; o < 64; o <<= 1) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) sm[k] += __shfl_xor(sm[k], o, 64);
                }
                float t = -q;
                if (grp == 0 && lane_on) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) t = fmaf(sm[k], sm[k], t);
                }
                fm = 0.5f * fx_wave_sum(t);
                if (a.S && grp == 0 && lane_on) fx_store<VEC>(a.S + b * a.D + d0, sm);
            }
            if (lane == 0) {
                if (a.lr_out) a.lr_out[b] = lr;
                if (a.fm_out) a.fm_out[b] = fm;
                if (a.fm_lr_out) a.fm_lr_out[b] = fm + lr;
            }
            fx_zero_reserved(a, b, lane);
        }
    }
}

extern "C" int fx_emb_fm_fwd(const void* table, int32_t table_dtype, int32_t D, const int32_t* ids,
                             int64_t ids_ld,
                             const int64_t* col_row_base, const int32_t* col_vocab,
                             const int64_t* col_out_off, int32_t C, const float* dense,
                             int64_t dense_ld, const float* num_w, const int64_t* num_out_off,
                             int32_t Fd, float* out, int64_t out_ld, int64_t B,
                             const float* table1, const float* num_w1, const float* bias1,
                             float* lr_out, float* fm_out, float* fm_lr_out, float* S,
                             fx_scalars* scal, int64_t table_ld, int64_t table1_ld,
                             int64_t zero_off0, int32_t zero_n0, int64_t zero_off1, int32_t zero_n1,
                             fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256, "fx_emb_fm_fwd: D=%d not in [1,256]", D);
    if (table_ld <= 0) table_ld = D;
    if (table1_ld <= 0) table1_ld = 1;
    FX_CHECK_ARG(table_dtype == FX_F32 || table_dtype == FX_BF16,
                 "fx_emb_fm_fwd: table_dtype must be FX_F32 or FX_BF16");
    FX_CHECK_ARG(C >= 0 && Fd >= 0 && B >= 0, "fx_emb_fm_fwd: negative size");
    if (B == 0 || C + Fd == 0) return FX_OK;
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(g.lanes <= 64, "fx_emb_fm_fwd: D=%d needs %d lanes per row (max 64)", D, g.lanes);
    FX_CHECK_ARG(out && scal, "fx_emb_fm_fwd: null out/scal");
    FX_CHECK_ARG(C == 0 || (table && ids && col_row_base && col_vocab && col_out_off),
                 "fx_emb_fm_fwd: null sparse argument");
    FX_CHECK_ARG(Fd == 0 || (dense && num_w && num_out_off), "fx_emb_fm_fwd: null numeric argument");
    const bool want_lr = lr_out != nullptr || fm_lr_out != nullptr;
    FX_CHECK_ARG(!want_lr || ((C == 0 || table1) && (Fd == 0 || num_w1)),
                 "fx_emb_fm_fwd: first-order term requested without its D=1 table / numeric weights");
    FX_CHECK_ARG(out_ld % g.vec == 0, "fx_emb_fm_fwd: out_ld=%lld not a multiple of %d",
                 (long long)out_ld, g.vec);
    FX_CHECK_ARG(table_ld >= D && table_ld % g.vec == 0 &&
                     (reinterpret_cast<uintptr_t>(table) % (g.vec * (table_dtype == FX_BF16 ? 2 : 4))) == 0,
                 "fx_emb_fm_fwd: table_ld=%lld / table alignment does not allow %d-wide row loads",
                 (long long)table_ld, g.vec);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    EmbFmArgs a{table, ids, ids_ld, col_row_base, col_vocab, col_out_off, dense, dense_ld, num_w,
                num_out_off, out, out_ld, B, table1, num_w1, bias1, lr_out, fm_out, fm_lr_out, S,
                scal, D, C, Fd, ll, table_dtype == FX_BF16 ? 1 : 0, table_ld, table1_ld,
                {zero_off0, zero_off1}, {zero_n0, zero_n1}};
    FX_CHECK_ARG(zero_n0 >= 0 && zero_n1 >= 0 && zero_off0 >= 0 && zero_off1 >= 0 &&
                     zero_off0 + zero_n0 <= out_ld && zero_off1 + zero_n1 <= out_ld,
                 "fx_emb_fm_fwd: reserved-slot range outside the record row");
    hipStream_t s = fx_hip_stream(stream);
    // A/B switch, read per call so that a test can compare the two forms in one process:
    // FX_EMB_FWD2=0 = the first version for every shape
    const char* e2 = getenv("FX_EMB_FWD2");
    const bool v2 = !(e2 && atoi(e2) == 0);
    const int ngrp = 64 / g.lanes;
    const int ni = (int)fx_ceil_div(C + Fd, ngrp);
    if (v2 && C <= 64 && Fd <= 64 && ni <= 4) {
        // two samples per wave once the batch alone fills the chip twice over (8192 resident waves)
        const int spw = B >= 16384 ? 2 : 1;
        int64_t blocks = fx_ceil_div(B, 4 * spw);
        if (blocks > 256 * 32) blocks = 256 * 32;
        dim3 grid((unsigned)blocks);
#define FX_FWD2(V, N)                                                                              \
        do {                                                                                       \
            if (spw == 2) hipLaunchKernelGGL((k_emb_fm_fwd2<V, N, 2>), grid, dim3(256), 0, s, a);  \
            else hipLaunchKernelGGL((k_emb_fm_fwd2<V, N, 1>), grid, dim3(256), 0, s, a);           \
        } while (0)
#define FX_FWD2_NI(V)                                                                              \
        do {                                                                                       \
            if (ni <= 1) FX_FWD2(V, 1); else if (ni == 2) FX_FWD2(V, 2);                           \
            else if (ni == 3) FX_FWD2(V, 3); else FX_FWD2(V, 4);                                   \
        } while (0)
        if (g.vec == 4) FX_FWD2_NI(4);
        else if (g.vec == 2) FX_FWD2_NI(2);
        else FX_FWD2_NI(1);
#undef FX_FWD2_NI
#undef FX_FWD2
        FX_CHECK_LAUNCH();
        return FX_OK;
    }
    int64_t blocks = fx_ceil_div(B, 4);
    if (blocks > 256 * 32) blocks = 256 * 32;
    dim3 grid((unsigned)blocks);
    if (g.vec == 4) hipLaunchKernelGGL(k_emb_fm_fwd<4>, grid, dim3(256), 0, s, a);
    else if (g.vec == 2) hipLaunchKernelGGL(k_emb_fm_fwd<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_emb_fm_fwd<1>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_emb_fm_bwd: gradient of every unique row, balanced over the SORTED LOOKUPS.
//
// Value of lookup (b,c):  drec[b, off_c + d]  +  g_fm[b] * (S[b,d] - rec[b, off_c + d])   [D-float row]
//                         g_lr[b]                                                          [D=1 row]
// the second term being d/de of 0.5 * sum_d((sum_f e)^2 - sum_f e^2) (inner_product.py:56-62).
//
// A row's lookups are a run of the sorted array; run lengths go from 1 to ~B (the hot row of a
// 3-row table holds 2800 of 4096 lookups under the power-law ids), so handing out ROWS leaves a few
// workgroups with thousands of dependent loads (first version: 280 us, the rest of the chip idle).
// Here workgroup k takes the sorted lookups [k*T, (k+1)*T), lane group g the T/NG lookups of piece
// g — every lane group of the launch issues the same FX_BWD_INFL independent row loads, once.
// A run that lies inside one piece is finished by its lane group (ascending order, as
// fx_emb_grad_reduce sums it); pieces of a longer run are combined in group order inside the
// workgroup (LDS), and the two open ends of a workgroup — the run that began in an earlier
// workgroup, the run that goes on into the next — are left as "edges" that launch 2 combines in
// workgroup order.  Every sum has a fixed order given the data layout: deterministic.
//
// Launch 1 also reduces the numeric features' partial sums over NC row chunks (extra workgroups);
// launch 2: edge combine + numeric finals (dnum_w, dnum_w1, dbias1).
// ---------------------------------------------------------------------------------------------
#define FX_BWD_T 256        // sorted lookups per workgroup of launch 1
#define FX_BWD_INFL 4       // lookups per lane group (= FX_BWD_T / NG at D = 16), all in flight
#define FX_BWD_NC 16        // row chunks of the numeric-gradient partial sums

struct EmbFmBwdArgs {
    const float* drec;
    int64_t drec_ld;
    const float* rec;
    int64_t rec_ld;
    const float* S;
    const float* g_fm;
    const float* g_lr;
    const int64_t* col_out_off;
    const uint32_t* sorted_pos;
    const uint32_t* sorted_uid;
    const uint32_t* seg_start;
    const int32_t* n_unique;
    float* G;
    float* sq_partials;
    float* G1;
    float* sq1_partials;
    // edges of launch-1 workgroups (workspace): per workgroup k
    float* edgeF;      // [nb, D]   partial of the run that began before the workgroup's range
    float* edgeL;      // [nb, D]   partial of the run that goes on past it
    float* edgeF1;     // [nb]      the same for the D=1 table
    float* edgeL1;
    int32_t* edge_row; // [nb, 3]   rowF, rowL (unique-row index or -1), wholeF (range = one run)
    // numeric features
    const float* dense;
    int64_t dense_ld;
    const int64_t* num_out_off;
    float* num_part;   // [NC, Fd*D + Fd + 1] partial sums (workspace)
    float* dnum_w;
    float* dnum_w1;
    float* dbias1;
    int64_t B, n;      // n = B*C sorted lookups
    int32_t C, D, Fd, lanes_log2, nb, n_num_blocks;
};

template <int VEC, bool FM, bool LR>
__device__ __forceinline__ void fx_lookup_value(const EmbFmBwdArgs& a, uint32_t p, int d0,
                                                bool lane_on, float (&v)[VEC], float& v1) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) v[k] = 0.f;
    v1 = 0.f;
    if (p == 0xFFFFFFFFu) return;            // padding_idx / bad-id lookup: contributes nothing
    const uint32_t b = p / (uint32_t)a.C, c = p - b * (uint32_t)a.C;
    if constexpr (LR) v1 = a.g_lr[b];
    if (!lane_on) return;
    const int64_t off = a.col_out_off[c] + d0;
    if (a.drec) fx_load<VEC>(a.drec + (int64_t)b * a.drec_ld + off, v);
    if constexpr (FM) {
        float e[VEC], s[VEC];
        fx_load<VEC>(a.rec + (int64_t)b * a.rec_ld + off, e);
        fx_load<VEC>(a.S + (int64_t)b * a.D + d0, s);
        const float g = a.g_fm[b];
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] += g * (s[k] - e[k]);
    }
}

// numeric partial sums: workgroup (j, chunk); j < Fd: feature j (and its D=1 twin), j == Fd: LR bias
template <bool FM, bool LR>
__device__ __forceinline__ void fx_numeric_partial(const EmbFmBwdArgs& a, int job, float* red) {
    const int nj = a.Fd + 1;
    const int j = job % nj, chunk = job / nj;
    int Dp = 1;
    while (Dp < a.D) Dp <<= 1;
    if (Dp > 256) Dp = 256;
    const int d = threadIdx.x % Dp, grp = threadIdx.x / Dp, ngrp = 256 / Dp;
    const int64_t rows = (a.B + FX_BWD_NC - 1) / FX_BWD_NC;
    const int64_t b0 = (int64_t)chunk * rows;
    const int64_t b1 = (b0 + rows < a.B) ? b0 + rows : a.B;
    float acc = 0.f, acc1 = 0.f;
    const int stride = a.Fd * a.D + a.Fd + 1;
    float* out = a.num_part + (int64_t)chunk * stride;
    if (j < a.Fd) {
        const int64_t off = a.num_out_off[j];
        for (int dd0 = 0; dd0 < a.D; dd0 += Dp) {                  // (one pass unless D > 256)
            const int dcol = dd0 + d;
            acc = 0.f;
            for (int64_t b = b0 + grp; b < b1; b += 8 * ngrp) {   // 8 independent rows in flight
                float x[8], v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t bb = b + (int64_t)u * ngrp;
                    x[u] = 0.f;
                    v[u] = 0.f;
                    if (bb < b1 && dcol < a.D) {
                        x[u] = a.dense[bb * a.dense_ld + j];
                        float t = a.drec ? a.drec[bb * a.drec_ld + off + dcol] : 0.f;
                        if constexpr (FM)
                            t += a.g_fm[bb] * (a.S[bb * a.D + dcol] - a.rec[bb * a.rec_ld + off + dcol]);
                        v[u] = t;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = fmaf(x[u], v[u], acc);
            }
            red[threadIdx.x] = acc;
            __syncthreads();
            for (int s = ngrp >> 1; s > 0; s >>= 1) {
                if (grp < s) red[threadIdx.x] += red[threadIdx.x + s * Dp];
                __syncthreads();
            }
            if (grp == 0 && dcol < a.D) out[(int64_t)j * a.D + dcol] = red[d];
            __syncthreads();
        }
        if constexpr (LR) {
            for (int64_t b = b0 + threadIdx.x; b < b1; b += 256)
                acc1 = fmaf(a.dense[b * a.dense_ld + j], a.g_lr[b], acc1);
        }
    } else {
        if constexpr (LR) {
            for (int64_t b = b0 + threadIdx.x; b < b1; b += 256) acc1 += a.g_lr[b];
        }
    }
    if constexpr (LR) {
        red[threadIdx.x] = acc1;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) out[(int64_t)a.Fd * a.D + j] = red[0];   // j == Fd: the bias slot
    }
}

template <int VEC, bool FM, bool LR>
__global__ __launch_bounds__(256) void k_emb_fm_bwd(EmbFmBwdArgs a) {
    // open pieces: F = the piece's first run began before the piece, L = its last run goes on
    __shared__ float openF[256 * VEC], openL[256 * VEC];
    __shared__ float openF1[256], openL1[256];
    __shared__ int32_t rowF[256], rowL[256], wholeF[256];     // per lane group; row = -1: none
    __shared__ float red[256];
    if ((int)blockIdx.x >= a.nb) {                               // block-uniform
        fx_numeric_partial<FM, LR>(a, (int)blockIdx.x - a.nb, red);
        return;
    }
    const int lanes = 1 << a.lanes_log2;
    const int NG = 256 >> a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int g = threadIdx.x >> a.lanes_log2;
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int64_t s0 = (int64_t)blockIdx.x * FX_BWD_T;
    const int64_t s1 = (s0 + FX_BWD_T < a.n) ? s0 + FX_BWD_T : a.n;
    const int64_t len = s1 - s0;
    const int64_t lo = s0 + (len * g) / NG, hi = s0 + (len * (g + 1)) / NG;
    float sq = 0.f, sq1 = 0.f;
    if (sub == 0) {
        rowF[g] = -1;
        rowL[g] = -1;
        wholeF[g] = 0;
    }
    if (lo < hi) {
        int64_t i = lo;
        while (i < hi) {
            const uint32_t u = a.sorted_uid[i];
            // generic de-dup results (fx_dedup, not the column path) put the padding / bad-id lookups
            // at the END of the sorted array with uid 0xFFFFFFFF: nothing follows them
            if (u == 0xFFFFFFFFu) break;
            const int64_t rbeg = a.seg_start[u], rend = a.seg_start[u + 1];
            const int64_t end = rend < hi ? rend : hi;
            float acc[VEC], acc1 = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
            for (int64_t i2 = i; i2 < end; i2 += FX_BWD_INFL) {
                uint32_t p[FX_BWD_INFL];
                float v[FX_BWD_INFL][VEC], v1[FX_BWD_INFL];
#pragma unroll
                for (int j = 0; j < FX_BWD_INFL; ++j)
                    p[j] = (i2 + j < end) ? a.sorted_pos[i2 + j] : 0xFFFFFFFFu;
#pragma unroll
                for (int j = 0; j < FX_BWD_INFL; ++j)
                    fx_lookup_value<VEC, FM, LR>(a, p[j], d0, lane_on, v[j], v1[j]);
#pragma unroll
                for (int j = 0; j < FX_BWD_INFL; ++j) {
#pragma unroll
                    for (int k = 0; k < VEC; ++k) acc[k] += v[j][k];
                    acc1 += v1[j];
                }
            }
            const bool began_before = rbeg < lo, goes_on = rend > hi;
            if (!began_before && !goes_on) {                     // the whole run: final
                if (lane_on) {
                    fx_store<VEC>(a.G + (int64_t)u * a.D + d0, acc);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) sq = fmaf(acc[k], acc[k], sq);
                }
                if constexpr (LR) {
                    if (sub == 0) {
                        a.G1[u] = acc1;
                        sq1 = fmaf(acc1, acc1, sq1);
                    }
                }
            } else if (began_before) {                           // tail (or middle) of a run
#pragma unroll
                for (int k = 0; k < VEC; ++k) openF[k * 256 + threadIdx.x] = acc[k];
                if (sub == 0) {
                    openF1[g] = acc1;
                    rowF[g] = (int32_t)u;
                    wholeF[g] = goes_on ? 1 : 0;
                }
            } else {                                             // head of a run that goes on
#pragma unroll
                for (int k = 0; k < VEC; ++k) openL[k * 256 + threadIdx.x] = acc[k];
                if (sub == 0) {
                    openL1[g] = acc1;
                    rowL[g] = (int32_t)u;
                }
            }
            i = end;
        }
    }
    __syncthreads();
    // chains inside the workgroup.  A chain starts at a head (rowL[g]) or, for the run that began
    // in an earlier workgroup, at lane group 0's openF; it absorbs the following pieces' openF of the
    // same run while they are "whole", and the first non-whole one ends it.
    const bool head = rowL[g] >= 0;
    // (the first NON-EMPTY piece: pieces in front of it are empty when the range is shorter than NG)
    const bool inherited = (lo == s0) && (lo < hi) && rowF[g] >= 0;
    if (sub == 0 && g == 0) {
        a.edge_row[3 * blockIdx.x + 0] = -1;
        a.edge_row[3 * blockIdx.x + 1] = -1;
        a.edge_row[3 * blockIdx.x + 2] = 0;
    }
    __syncthreads();
    for (int pass = 0; pass < 2; ++pass) {
        const bool mine = pass == 0 ? inherited : head;
        if (!mine) continue;
        const int r = pass == 0 ? rowF[g] : rowL[g];
        float tot[VEC], tot1;
        bool ended;
        if (pass == 0) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] = openF[k * 256 + threadIdx.x];
            tot1 = openF1[g];
            ended = !wholeF[g];
        } else {
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] = openL[k * 256 + threadIdx.x];
            tot1 = openL1[g];
            ended = false;
        }
        for (int h = g + 1; h < NG && !ended; ++h) {
            if (rowF[h] != r) {
                if (rowF[h] < 0 && rowL[h] < 0) continue;          // an empty piece (len < NG)
                break;                                             // (cannot happen: runs are contiguous)
            }
            const int th = (h << a.lanes_log2) + sub;
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] += openF[k * 256 + th];
            tot1 += openF1[h];
            ended = !wholeF[h];
        }
        if (pass == 1 && ended) {                                // began and ended in this workgroup
            if (lane_on) {
                fx_store<VEC>(a.G + (int64_t)r * a.D + d0, tot);
#pragma unroll
                for (int k = 0; k < VEC; ++k) sq = fmaf(tot[k], tot[k], sq);
            }
            if constexpr (LR) {
                if (sub == 0) {
                    a.G1[r] = tot1;
                    sq1 = fmaf(tot1, tot1, sq1);
                }
            }
        } else {                                                 // an edge for launch 2
            float* e = (pass == 0 ? a.edgeF : a.edgeL) + (int64_t)blockIdx.x * a.D;
            if (lane_on) fx_store<VEC>(e + d0, tot);
            if (sub == 0) {
                if constexpr (LR) (pass == 0 ? a.edgeF1 : a.edgeL1)[blockIdx.x] = tot1;
                a.edge_row[3 * blockIdx.x + pass] = r;
                if (pass == 0) a.edge_row[3 * blockIdx.x + 2] = ended ? 0 : 1;
            }
        }
    }
    // ||G||^2 of the rows finished here, fixed order
    __syncthreads();
    {
        float* red4 = red;
        const float tot = fx_block_sum_256(sq, red4);
        if (threadIdx.x == 0) a.sq_partials[blockIdx.x] = tot;
        if constexpr (LR) {
            __syncthreads();
            const float t1 = fx_block_sum_256(sq1, red4);
            if (threadIdx.x == 0) a.sq1_partials[blockIdx.x] = t1;
        }
    }
}

// launch 2.  Workgroups [0, ncomb): lane group q combines the run whose head is the edgeL of launch-1
// workgroup q (+ the edgeF of the workgroups after it); last workgroup: numeric finals.
template <int VEC, bool LR>
__global__ __launch_bounds__(256) void k_emb_fm_bwd_finish(EmbFmBwdArgs a, int ncomb) {
    __shared__ float red4[4];
    if ((int)blockIdx.x >= ncomb) {                              // numeric finals (block-uniform)
        const int stride = a.Fd * a.D + a.Fd + 1;
        for (int t = threadIdx.x; t < stride; t += 256) {
            float s = 0.f;
            for (int c = 0; c < FX_BWD_NC; ++c) s += a.num_part[(int64_t)c * stride + t];
            if (t < a.Fd * a.D) { if (a.dnum_w) a.dnum_w[t] = s; }
            else if (t < a.Fd * a.D + a.Fd) { if (LR && a.dnum_w1) a.dnum_w1[t - a.Fd * a.D] = s; }
            else { if (LR && a.dbias1) a.dbias1[0] = s; }
        }
        return;
    }
    const int lanes = 1 << a.lanes_log2;
    const int NG = 256 >> a.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int d0 = sub * VEC;
    const bool lane_on = d0 < a.D;
    const int q = (int)blockIdx.x * NG + (threadIdx.x >> a.lanes_log2);
    float sq = 0.f, sq1 = 0.f;
    if (q < a.nb) {
        const int r = a.edge_row[3 * q + 1];
        if (r >= 0) {
            float tot[VEC], tot1 = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) tot[k] = 0.f;
            if (lane_on) fx_load<VEC>(a.edgeL + (int64_t)q * a.D + d0, tot);
            if constexpr (LR) tot1 = a.edgeL1[q];
            for (int h = q + 1; h < a.nb; ++h) {
                if (a.edge_row[3 * h + 0] != r) break;
                if (lane_on) {
                    float e[VEC];
                    fx_load<VEC>(a.edgeF + (int64_t)h * a.D + d0, e);
#pragma unroll
                    for (int k = 0; k < VEC; ++k) tot[k] += e[k];
                }
                if constexpr (LR) tot1 += a.edgeF1[h];
                if (!a.edge_row[3 * h + 2]) break;
            }
            if (lane_on) {
                fx_store<VEC>(a.G + (int64_t)r * a.D + d0, tot);
#pragma unroll
                for (int k = 0; k < VEC; ++k) sq = fmaf(tot[k], tot[k], sq);
            }
            if constexpr (LR) {
                if (sub == 0) {
                    a.G1[r] = tot1;
                    sq1 = fmaf(tot1, tot1, sq1);
                }
            }
        }
    }
    const float tot = fx_block_sum_256(sq, red4);
    if (threadIdx.x == 0) a.sq_partials[a.nb + blockIdx.x] = tot;
    if constexpr (LR) {
        __syncthreads();
        const float t1 = fx_block_sum_256(sq1, red4);
        if (threadIdx.x == 0) a.sq1_partials[a.nb + blockIdx.x] = t1;
    }
}

static inline int64_t fx_bwd_nb(int64_t n_lookups) { return fx_ceil_div(n_lookups > 0 ? n_lookups : 1, FX_BWD_T); }
static inline int64_t fx_bwd_ncomb(int64_t n_lookups, int32_t D) {
    return fx_ceil_div(fx_bwd_nb(n_lookups), 256 / fx_row_geom(D).lanes);
}

extern "C" int64_t fx_emb_fm_bwd_partials(int64_t n_lookups, int32_t D) {
    if (D < 1 || D > 256) return 1;
    return fx_bwd_nb(n_lookups) + fx_bwd_ncomb(n_lookups, D);
}

extern "C" int64_t fx_emb_fm_bwd_workspace_floats(int64_t n_lookups, int32_t D, int32_t Fd) {
    if (D < 1 || D > 256 || Fd < 0) return 0;
    const int64_t nb = fx_bwd_nb(n_lookups);
    return nb * (2 * (int64_t)D + 2 + 3) + (int64_t)FX_BWD_NC * ((int64_t)Fd * D + Fd + 1) + 16;
}

extern "C" int fx_emb_fm_bwd(const float* drec, int64_t drec_ld, const float* rec, int64_t rec_ld,
                             const float* S, const float* g_fm, const float* g_lr,
                             const int64_t* col_out_off, int32_t C, int32_t D,
                             const uint32_t* sorted_pos, const uint32_t* sorted_uid,
                             const uint32_t* seg_start, const int32_t* n_unique, int64_t n_max,
                             float* G, float* sq_partials, float* G1, float* sq1_partials,
                             const float* dense, int64_t dense_ld, const int64_t* num_out_off,
                             int32_t Fd, int64_t B, float* dnum_w, float* dnum_w1, float* dbias1,
                             float* workspace, fx_stream_t stream) {
    FX_CHECK_ARG(D >= 1 && D <= 256 && C >= 0 && Fd >= 0 && B >= 0, "fx_emb_fm_bwd: bad sizes");
    const FxRowGeom g = fx_row_geom(D);
    FX_CHECK_ARG(g.lanes <= 64, "fx_emb_fm_bwd: D=%d needs %d lanes per row (max 64)", D, g.lanes);
    FX_CHECK_ARG(g_fm == nullptr || (rec && S), "fx_emb_fm_bwd: FM term without rec / S");
    FX_CHECK_ARG(drec != nullptr || g_fm != nullptr || g_lr != nullptr,
                 "fx_emb_fm_bwd: no upstream gradient at all");
    FX_CHECK_ARG((drec == nullptr || drec_ld % g.vec == 0) && (rec == nullptr || rec_ld % g.vec == 0),
                 "fx_emb_fm_bwd: leading dimensions not a multiple of %d", g.vec);
    FX_CHECK_ARG(workspace, "fx_emb_fm_bwd: null workspace");
    const bool sparse = C > 0 && n_max > 0;
    const bool numeric = Fd > 0 || (g_lr && dbias1);
    if (!sparse && !numeric) return FX_OK;
    if (sparse) {
        FX_CHECK_ARG(col_out_off && sorted_pos && sorted_uid && seg_start && n_unique && G &&
                         sq_partials, "fx_emb_fm_bwd: null sparse argument");
        FX_CHECK_ARG(g_lr == nullptr || (G1 && sq1_partials), "fx_emb_fm_bwd: g_lr without G1");
        FX_CHECK_ARG(n_max == B * (int64_t)C, "fx_emb_fm_bwd: n_max must be B*C (one entry per lookup)");
    }
    FX_CHECK_ARG(!numeric || Fd == 0 || (dense && num_out_off && dnum_w),
                 "fx_emb_fm_bwd: null numeric argument");
    hipStream_t s = fx_hip_stream(stream);
    int ll = 0;
    while ((1 << ll) < g.lanes) ++ll;
    EmbFmBwdArgs a;
    memset(&a, 0, sizeof(a));
    a.drec = drec; a.drec_ld = drec_ld; a.rec = rec; a.rec_ld = rec_ld; a.S = S;
    a.g_fm = g_fm; a.g_lr = g_lr; a.col_out_off = col_out_off;
    a.sorted_pos = sorted_pos; a.sorted_uid = sorted_uid; a.seg_start = seg_start;
    a.n_unique = n_unique; a.G = G; a.sq_partials = sq_partials; a.G1 = G1;
    a.sq1_partials = sq1_partials;
    a.dense = dense; a.dense_ld = dense_ld; a.num_out_off = num_out_off;
    a.dnum_w = dnum_w; a.dnum_w1 = dnum_w1; a.dbias1 = dbias1;
    a.B = B; a.n = sparse ? n_max : 0; a.C = C; a.D = D; a.Fd = Fd; a.lanes_log2 = ll;
    const int nb = sparse ? (int)fx_bwd_nb(n_max) : 0;
    const int ncomb = sparse ? (int)fx_bwd_ncomb(n_max, D) : 0;
    a.nb = nb;
    a.n_num_blocks = numeric ? (Fd + 1) * FX_BWD_NC : 0;
    // workspace carve (floats): edgeF | edgeL | edgeF1 | edgeL1 | edge_row (ints) | num_part
    float* w = workspace;
    const int64_t nbw = fx_bwd_nb(n_max > 0 ? n_max : 1);
    a.edgeF = w; w += nbw * D;
    a.edgeL = w; w += nbw * D;
    a.edgeF1 = w; w += nbw;
    a.edgeL1 = w; w += nbw;
    a.edge_row = reinterpret_cast<int32_t*>(w); w += nbw * 3;
    w = reinterpret_cast<float*>((reinterpret_cast<uintptr_t>(w) + 15) & ~(uintptr_t)15);
    a.num_part = w;
    dim3 grid1((unsigned)(nb + a.n_num_blocks)), grid2((unsigned)(ncomb + (numeric ? 1 : 0)));
#define FX_BWD_LAUNCH(V)                                                                          \
    do {                                                                                          \
        if (g_fm && g_lr) {                                                                       \
            hipLaunchKernelGGL((k_emb_fm_bwd<V, true, true>), grid1, dim3(256), 0, s, a);         \
            hipLaunchKernelGGL((k_emb_fm_bwd_finish<V, true>), grid2, dim3(256), 0, s, a, ncomb); \
        } else if (g_fm) {                                                                        \
            hipLaunchKernelGGL((k_emb_fm_bwd<V, true, false>), grid1, dim3(256), 0, s, a);        \
            hipLaunchKernelGGL((k_emb_fm_bwd_finish<V, false>), grid2, dim3(256), 0, s, a, ncomb); \
        } else if (g_lr) {                                                                        \
            hipLaunchKernelGGL((k_emb_fm_bwd<V, false, true>), grid1, dim3(256), 0, s, a);        \
            hipLaunchKernelGGL((k_emb_fm_bwd_finish<V, true>), grid2, dim3(256), 0, s, a, ncomb); \
        } else {                                                                                  \
            hipLaunchKernelGGL((k_emb_fm_bwd<V, false, false>), grid1, dim3(256), 0, s, a);       \
            hipLaunchKernelGGL((k_emb_fm_bwd_finish<V, false>), grid2, dim3(256), 0, s, a, ncomb); \
        }                                                                                         \
    } while (0)
    if (g.vec == 4) FX_BWD_LAUNCH(4);
    else if (g.vec == 2) FX_BWD_LAUNCH(2);
    else FX_BWD_LAUNCH(1);
#undef FX_BWD_LAUNCH
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_sparse_adam_multi / fx_sparse_sgd_multi: the row update of every table group that shares one
// de-dup result (same uniq_row), one launch.
// ---------------------------------------------------------------------------------------------
struct MultiOptArgs {
    FxTableDev t[FX_MAX_TABLES];
    const uint32_t* uniq_row;
    const int32_t* n_unique;
    const fx_scalars* scal;
    int32_t n_tables, group_log2;
};

__device__ __forceinline__ float fx_reg_grad2(float p, float l1, float l2) {
    float r = l2 * p;
    if (l1 != 0.f) r += p > 0.f ? l1 : (p < 0.f ? -l1 : 0.f);
    return r;
}

template <int VEC, bool ADAM>
__device__ __forceinline__ void fx_update_row(const FxTableDev& t, int64_t u, int64_t row, int sub,
                                              const fx_scalars& sc) {
    const int lanes = 1 << t.lanes_log2;
    if (sub >= lanes) return;
    const int d0 = sub * VEC;
    if (d0 < t.D) {
        float p[VEC], g[VEC];
        const int64_t o = row * t.tld + d0;
        fx_tab_load<VEC>(t.table, t.bf16, o, p);
        fx_load<VEC>(t.G + u * t.D + d0, g);
        if (sc.reg_l1 != 0.f || sc.reg_l2 != 0.f) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] += fx_reg_grad2(p[k], sc.reg_l1, sc.reg_l2);
        }
        if constexpr (ADAM) {
            float m[VEC], v[VEC];
            const int64_t om = row * t.mld + d0, ov = row * t.vld + d0;
            fx_load<VEC>(t.m + om, m);
            fx_load<VEC>(t.v + ov, v);
            const float w1 = fx_one_minus(sc.beta1), w2 = fx_one_minus(sc.beta2);   // torch's float(1 - beta)
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                const float gk = g[k] * sc.clip_coef;
                m[k] = m[k] + w1 * (gk - m[k]);                      // exp_avg.lerp_(grad, 1 - beta1)
                v[k] = fmaf(w2 * gk, gk, v[k] * sc.beta2);           // exp_avg_sq ... addcmul_
                const float denom = sqrtf(v[k]) / sc.bc2_sqrt + sc.eps;
                p[k] = p[k] - sc.step_size * (m[k] / denom);         // param.addcdiv_
            }
            fx_store<VEC>(t.m + om, m);
            fx_store<VEC>(t.v + ov, v);
        } else {
            const float scale = sc.lr * sc.clip_coef;
#pragma unroll
            for (int k = 0; k < VEC; ++k) p[k] = p[k] - scale * g[k];
        }
        fx_tab_store<VEC>(t.table, t.bf16, o, p);
    }
    if (sub == 0 && t.last_step) t.last_step[row * t.lld] = sc.step;
}

// the Adam half of fx_update_row on registers that are already loaded
template <int VEC>
__device__ __forceinline__ void fx_adam_finish(const FxTableDev& t, int64_t row, int sub,
                                               FxRowRegs<VEC>& r, float (&g)[VEC],
                                               const fx_scalars& sc) {
    if (!r.act) return;
    if (r.on) {
        if (sc.reg_l1 != 0.f || sc.reg_l2 != 0.f) {
#pragma unroll
            for (int k = 0; k < VEC; ++k) g[k] += fx_reg_grad2(r.p[k], sc.reg_l1, sc.reg_l2);
        }
        const float w1 = fx_one_minus(sc.beta1), w2 = fx_one_minus(sc.beta2);   // torch's float(1 - beta)
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const float gk = g[k] * sc.clip_coef;
            r.m[k] = r.m[k] + w1 * (gk - r.m[k]);
            r.v[k] = fmaf(w2 * gk, gk, r.v[k] * sc.beta2);
            const float denom = sqrtf(r.v[k]) / sc.bc2_sqrt + sc.eps;
            r.p[k] = r.p[k] - sc.step_size * (r.m[k] / denom);
        }
        fx_tab_store<VEC>(t.table, t.bf16, row * t.tld + sub * VEC, r.p);
        fx_store<VEC>(t.m + row * t.mld + sub * VEC, r.m);
        fx_store<VEC>(t.v + row * t.vld + sub * VEC, r.v);
    }
    if (sub == 0 && t.last_step) t.last_step[row * t.lld] = sc.step;
}

template <bool ADAM>
__global__ __launch_bounds__(256) void k_sparse_update_multi(MultiOptArgs a) {
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t rpb = 256 >> a.group_log2;
    const int nu = *a.n_unique;
    const fx_scalars sc = *a.scal;
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.group_log2); u < nu;
         u += (int64_t)gridDim.x * rpb) {
        const int64_t row = a.uniq_row[u];
        if (ADAM && a.n_tables == 2 && a.t[0].vec == 4 && a.t[1].vec == 1) {
            FxRowRegs<4> r0;
            FxRowRegs<1> r1;
            float g0[4] = {0.f, 0.f, 0.f, 0.f}, g1[1] = {0.f};
            fx_row_load<4, false>(a.t[0], row, sub, r0);
            fx_row_load<1, false>(a.t[1], row, sub, r1);
            if (r0.on) fx_load<4>(a.t[0].G + u * a.t[0].D + sub * 4, g0);
            if (r1.on) fx_load<1>(a.t[1].G + u * a.t[1].D + sub, g1);
            fx_adam_finish<4>(a.t[0], row, sub, r0, g0, sc);
            fx_adam_finish<1>(a.t[1], row, sub, r1, g1, sc);
            continue;
        }
        for (int t = 0; t < a.n_tables; ++t) {
            const FxTableDev& tb = a.t[t];
            if (tb.vec == 4) fx_update_row<4, ADAM>(tb, u, row, sub, sc);
            else if (tb.vec == 2) fx_update_row<2, ADAM>(tb, u, row, sub, sc);
            else fx_update_row<1, ADAM>(tb, u, row, sub, sc);
        }
    }
}

static int fx_sparse_update_multi(bool adam, const fx_row_state* tables_host, int32_t n_tables,
                                  const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                                  const fx_scalars* scal, fx_stream_t stream, const char* who) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_MAX_TABLES, "%s: n_tables=%d not in [1,%d]", who,
                 n_tables, FX_MAX_TABLES);
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(tables_host && uniq_row && n_unique && scal, "%s: null pointer", who);
    MultiOptArgs a;
    memset(&a, 0, sizeof(a));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, a.t, &gl, who, false);
    if (st != FX_OK) return st;
    for (int t = 0; t < n_tables; ++t) {
        FX_CHECK_ARG(a.t[t].G != nullptr, "%s: table %d has no gradient", who, t);
        FX_CHECK_ARG(!adam || (a.t[t].m && a.t[t].v), "%s: table %d has no Adam moments", who, t);
    }
    a.uniq_row = uniq_row;
    a.n_unique = n_unique;
    a.scal = scal;
    a.n_tables = n_tables;
    a.group_log2 = gl;
    int64_t blocks = fx_ceil_div(n_max, 256 >> gl);
    if (blocks > 256 * 64) blocks = 256 * 64;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (adam) hipLaunchKernelGGL(k_sparse_update_multi<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(k_sparse_update_multi<false>, grid, dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_sparse_adam_multi(const fx_row_state* tables_host, int32_t n_tables,
                                    const uint32_t* uniq_row, const int32_t* n_unique,
                                    int64_t n_max, const fx_scalars* scal, fx_stream_t stream) {
    return fx_sparse_update_multi(true, tables_host, n_tables, uniq_row, n_unique, n_max, scal,
                                  stream, "fx_sparse_adam_multi");
}

extern "C" int fx_sparse_sgd_multi(const fx_row_state* tables_host, int32_t n_tables,
                                   const uint32_t* uniq_row, const int32_t* n_unique, int64_t n_max,
                                   const fx_scalars* scal, fx_stream_t stream) {
    return fx_sparse_update_multi(false, tables_host, n_tables, uniq_row, n_unique, n_max, scal,
                                  stream, "fx_sparse_sgd_multi");
}

// ---------------------------------------------------------------------------------------------
// fx_adam_catchup_all: flush of the exact mode for fp32 or bf16 tables — every row of the table is
// brought up to step + upto_offset (before evaluate / save / a learning-rate change).
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void k_catchup_all(FxTableDev t, int64_t total_rows,
                                                     const fx_scalars* scal, int upto_offset) {
    const int lanes = 1 << t.lanes_log2;
    const int sub = threadIdx.x & (lanes - 1);
    const int64_t rpb = 256 >> t.lanes_log2;
    const fx_scalars sc = *scal;
    const int upto = sc.step + upto_offset;
    const FxLogs lg = fx_logs_of(sc);
    const FxSeries ser = fx_series_of(scal, sc);
    for (int64_t row = (int64_t)blockIdx.x * rpb + (threadIdx.x >> t.lanes_log2); row < total_rows;
         row += (int64_t)gridDim.x * rpb) {
        if (t.last_step[row * t.lld] >= upto) continue;
        fx_catchup_row<VEC>(t, row, sub, sc, upto, lg, ser);
    }
}

extern "C" int fx_adam_catchup_all(const fx_row_state* table_host, int64_t total_rows,
                                   int32_t upto_offset, const fx_scalars* scal,
                                   fx_stream_t stream) {
    FX_CHECK_ARG(table_host && scal, "fx_adam_catchup_all: null pointer");
    if (total_rows <= 0) return FX_OK;
    FxTableDev t;
    int gl = 0;
    const int st = fx_fill_tables(table_host, 1, &t, &gl, "fx_adam_catchup_all", true);
    if (st != FX_OK) return st;
    int64_t blocks = fx_ceil_div(total_rows, 256 >> t.lanes_log2);
    if (blocks > 256 * 64) blocks = 256 * 64;
    dim3 grid((unsigned)blocks);
    hipStream_t s = fx_hip_stream(stream);
    if (t.vec == 4) hipLaunchKernelGGL(k_catchup_all<4>, grid, dim3(256), 0, s, t, total_rows, scal, (int)upto_offset);
    else if (t.vec == 2) hipLaunchKernelGGL(k_catchup_all<2>, grid, dim3(256), 0, s, t, total_rows, scal, (int)upto_offset);
    else hipLaunchKernelGGL(k_catchup_all<1>, grid, dim3(256), 0, s, t, total_rows, scal, (int)upto_offset);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_adam_catchup_rows: the exact-mode catch-up of the unique rows of ANY de-dup result (the generic
// sort path: sequence columns that alias a table, batches beyond the column fast path), for fp32 or
// bf16 tables, every table group that shares the id plan in ONE launch.  (fx_adam_catchup is the
// fp32-only, one-table kernel of round 1; it must never see a bf16 table.)
// ---------------------------------------------------------------------------------------------
struct CatchRowsArgs {
    FxTableDev t[FX_MAX_TABLES];
    const uint32_t* uniq_row;
    const int32_t* n_unique;
    const fx_scalars* scal;
    int32_t n_tables, group_log2, upto_offset, quad;
};

__global__ __launch_bounds__(256) void k_catchup_rows(CatchRowsArgs a) {
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t rpb = 256 >> a.group_log2;
    const int nu = *a.n_unique;
    const fx_scalars sc = *a.scal;
    const int upto = sc.step + a.upto_offset;
    const FxLogs lg = fx_logs_of(sc);
    const FxSeries ser = fx_series_of(a.scal, sc);
    if (a.quad) {
        FxRowRegs<4> r0;
        FxRowRegs<1> r1;
        // (the votes inside fx_catchup_quad run over the lanes that are in it: quads past the end of the list
        // simply are not)
        if (a.quad == 1)
            for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> 2); u < nu; u += (int64_t)gridDim.x * rpb)
                fx_catchup_quad<true>(a.t[0], a.t[1], (int64_t)a.uniq_row[u], sub, sc, upto, lg, ser, r0, r1);
        else
            for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> 2); u < nu; u += (int64_t)gridDim.x * rpb)
                fx_catchup_quad<false>(a.t[0], a.t[0], (int64_t)a.uniq_row[u], sub, sc, upto, lg, ser, r0, r1);
        return;
    }
    for (int64_t u = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.group_log2); u < nu;
         u += (int64_t)gridDim.x * rpb)
        fx_catchup_tables(a.t, a.n_tables, (int64_t)a.uniq_row[u], sub, sc, upto, lg, ser);
}

static int fx_launch_catchup_rows(const FxTableDev* t, int n_tables, int gl, const uint32_t* uniq_row,
                                  const int32_t* n_unique, int64_t n_max, int32_t upto_offset,
                                  const fx_scalars* scal, hipStream_t s) {
    CatchRowsArgs a;
    memset(&a, 0, sizeof(a));
    for (int i = 0; i < n_tables; ++i) a.t[i] = t[i];
    a.uniq_row = uniq_row;
    a.n_unique = n_unique;
    a.scal = scal;
    a.n_tables = n_tables;
    a.group_log2 = gl;
    a.upto_offset = upto_offset;
    a.quad = !fx_catchup_quad_on || gl != 2 || t[0].vec != 4 || t[0].D != 16 ? 0
             : (n_tables == 2 && t[1].vec == 1 && t[1].D == 1) ? 1 : n_tables == 1 ? 2 : 0;
    int64_t blocks = fx_ceil_div(n_max, 256 >> gl);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_catchup_rows, dim3((unsigned)blocks), dim3(256), 0, s, a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

extern "C" int fx_adam_catchup_rows(const fx_row_state* tables_host, int32_t n_tables,
                                    const uint32_t* uniq_row, const int32_t* n_unique,
                                    int64_t n_max, int32_t upto_offset, const fx_scalars* scal,
                                    fx_stream_t stream) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_MAX_TABLES,
                 "fx_adam_catchup_rows: n_tables=%d not in [1,%d]", n_tables, FX_MAX_TABLES);
    if (n_max <= 0) return FX_OK;
    FX_CHECK_ARG(tables_host && uniq_row && n_unique && scal, "fx_adam_catchup_rows: null pointer");
    CatchRowsArgs a;
    memset(&a, 0, sizeof(a));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, a.t, &gl, "fx_adam_catchup_rows", true);
    if (st != FX_OK) return st;
    a.uniq_row = uniq_row;
    a.n_unique = n_unique;
    a.scal = scal;
    a.n_tables = n_tables;
    a.group_log2 = gl;
    a.upto_offset = upto_offset;
    a.quad = !fx_catchup_quad_on || gl != 2 || a.t[0].vec != 4 || a.t[0].D != 16 ? 0
             : (n_tables == 2 && a.t[1].vec == 1 && a.t[1].D == 1) ? 1 : n_tables == 1 ? 2 : 0;
    int64_t blocks = fx_ceil_div(n_max, 256 >> gl);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_catchup_rows, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_owner_fetch_rows: owner side of the row-sharded forward, ONE launch for every table group of the
// exchange (round 2: a catch-up launch + a gather launch per group).  A lane group owns one unique
// owned row of the de-dup of what the peers asked for: it brings the row up to date (exact mode: the
// zero-gradient Adam replay of fx_adam_catchup_rows, `catchup` != 0) and writes the row into the send
// block at EVERY position that asked for it (sorted_pos of its run: one entry per requesting rank), each
// group in its own columns [off_t, off_t + D_t) of the [n_total, ld] block; pad columns and the
// entries that asked for nothing (pad ids: the tail of the sorted array) are zeroed.  The rows are read
// once — the round-2 sequence read them in the catch-up and again in the gather.
// ---------------------------------------------------------------------------------------------
struct OwnerFetchArgs {
    FxTableDev t[FX_MAX_TABLES];
    int32_t off[FX_MAX_TABLES];
    const uint32_t* uniq_row;
    const uint32_t* seg_start;
    const uint32_t* sorted_pos;
    const int32_t* n_unique;
    float* send;
    int64_t ld, n_total;
    const fx_scalars* scal;
    float* zero_row;             // zero_w floats cleared on the way (pad row of the block the rows land in)
    int32_t n_tables, group_log2, upto_offset, catchup, used_w, zero_w, quad;
};

template <int VEC>
__device__ __forceinline__ void fx_owner_put(const OwnerFetchArgs& a, int ti, const FxRowRegs<VEC>& r,
                                             int sub, uint32_t beg, uint32_t end) {
    if (!r.on) return;
    float v[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k)
        // a bf16 table holds the ROUNDED row: that is what an unsharded gather would read back after the
        // catch-up stored it, so that is what travels (the registers still hold the unrounded fp32 result)
        v[k] = a.t[ti].bf16 ? fx_bf16_to_f32(fx_f32_to_bf16(r.p[k])) : r.p[k];
    for (uint32_t i = beg; i < end; ++i)
        fx_store<VEC>(a.send + (int64_t)a.sorted_pos[i] * a.ld + a.off[ti] + sub * VEC, v);
}

template <int VEC>
__device__ __forceinline__ void fx_owner_one(const OwnerFetchArgs& a, int ti, int64_t row, int sub,
                                             const fx_scalars& sc, int upto, const FxLogs& lg,
                                             const FxSeries& ser, uint32_t beg, uint32_t end) {
    FxRowRegs<VEC> r;
    if (a.catchup) {
        fx_row_load<VEC>(a.t[ti], row, sub, r);
        fx_catchup_finish<VEC>(a.t[ti], row, sub, r, sc, upto, lg, ser);
    } else {                                   // plain gather: only the row itself (no optimizer state)
        const int lanes = 1 << a.t[ti].lanes_log2;
        r.act = sub < lanes;
        r.on = r.act && sub * VEC < a.t[ti].D;
        if (r.on) fx_tab_load<VEC>(a.t[ti].table, a.t[ti].bf16, row * a.t[ti].tld + sub * VEC, r.p);
    }
    fx_owner_put<VEC>(a, ti, r, sub, beg, end);
}

__global__ __launch_bounds__(256) void k_owner_fetch_rows(OwnerFetchArgs a) {
    const int glanes = 1 << a.group_log2;
    const int sub = threadIdx.x & (glanes - 1);
    const int64_t rpb = 256 >> a.group_log2;
    const int nu = *a.n_unique;
    fx_scalars sc;
    int upto = 0;
    FxLogs lg{0.0, 0.0, 0.0, 0.0};
    FxSeries ser{nullptr, 0};
    if (a.catchup) {
        sc = *a.scal;
        upto = sc.step + a.upto_offset;
        lg = fx_logs_of(sc);
        ser = fx_series_of(a.scal, sc);
    }
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < a.zero_w; i += 256) a.zero_row[i] = 0.f;
    const int64_t gid = (int64_t)blockIdx.x * rpb + (threadIdx.x >> a.group_log2);
    const int64_t gstride = (int64_t)gridDim.x * rpb;
    for (int64_t u = gid; u < nu; u += gstride) {
        const int64_t row = a.uniq_row[u];
        const uint32_t beg = a.seg_start[u], end = a.seg_start[u + 1];
        if (a.quad) {
            // the shape of k_catchup_rows' quad replay, by the same function: a row is caught up to the same
            // bits whether its owner is this rank of N or the only rank (round 6; the plain replays below
            // round the k <= FX_SERIES_KDIR steps differently)
            FxRowRegs<4> r0;
            FxRowRegs<1> r1;
            if (a.quad == 1) fx_catchup_quad<true>(a.t[0], a.t[1], row, sub, sc, upto, lg, ser, r0, r1);
            else fx_catchup_quad<false>(a.t[0], a.t[0], row, sub, sc, upto, lg, ser, r0, r1);
            fx_owner_put<4>(a, 0, r0, sub, beg, end);
            if (a.quad == 1) fx_owner_put<1>(a, 1, r1, sub, beg, end);
        } else if (a.catchup && a.n_tables == 2 && a.t[0].vec == 4 && a.t[1].vec == 1) {
            // the D-float tables + the D=1 tables of LogisticRegression: all eight loads in flight
            FxRowRegs<4> r0;
            FxRowRegs<1> r1;
            fx_row_load<4>(a.t[0], row, sub, r0);
            fx_row_load<1>(a.t[1], row, sub, r1);
            fx_catchup_finish<4>(a.t[0], row, sub, r0, sc, upto, lg, ser);
            fx_catchup_finish<1>(a.t[1], row, sub, r1, sc, upto, lg, ser);
            fx_owner_put<4>(a, 0, r0, sub, beg, end);
            fx_owner_put<1>(a, 1, r1, sub, beg, end);
        } else {
            for (int ti = 0; ti < a.n_tables; ++ti) {
                const int vec = a.t[ti].vec;
                if (vec == 4) fx_owner_one<4>(a, ti, row, sub, sc, upto, lg, ser, beg, end);
                else if (vec == 2) fx_owner_one<2>(a, ti, row, sub, sc, upto, lg, ser, beg, end);
                else fx_owner_one<1>(a, ti, row, sub, sc, upto, lg, ser, beg, end);
            }
        }
        if (sub == 0)                                          // pad columns of the block's rows
            for (uint32_t i = beg; i < end; ++i)
                for (int c = a.used_w; c < (int)a.ld; ++c)
                    a.send[(int64_t)a.sorted_pos[i] * a.ld + c] = 0.f;
    }
    // entries that asked for nothing (pad ids sort to the tail): zero rows
    const int64_t n_valid = a.seg_start[nu];
    for (int64_t i = n_valid + (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n_total;
         i += (int64_t)gridDim.x * 256) {
        float* dst = a.send + (int64_t)a.sorted_pos[i] * a.ld;
        for (int c = 0; c < (int)a.ld; ++c) dst[c] = 0.f;
    }
}

extern "C" int fx_owner_fetch_rows(const fx_row_state* tables_host, const int32_t* off_host,
                                   int32_t n_tables, const uint32_t* uniq_row,
                                   const uint32_t* seg_start, const uint32_t* sorted_pos,
                                   const int32_t* n_unique, int64_t n_total, float* send, int64_t ld,
                                   int32_t catchup, int32_t upto_offset, const fx_scalars* scal,
                                   float* zero_row, int32_t zero_w, fx_stream_t stream) {
    FX_CHECK_ARG(n_tables >= 1 && n_tables <= FX_MAX_TABLES,
                 "fx_owner_fetch_rows: n_tables=%d not in [1,%d]", n_tables, FX_MAX_TABLES);
    FX_CHECK_ARG(zero_w >= 0 && (zero_w == 0 || zero_row), "fx_owner_fetch_rows: zero_w without zero_row");
    if (n_total <= 0) return FX_OK;
    FX_CHECK_ARG(tables_host && off_host && uniq_row && seg_start && sorted_pos && n_unique && send,
                 "fx_owner_fetch_rows: null pointer");
    FX_CHECK_ARG(!catchup || scal, "fx_owner_fetch_rows: catch-up without scal");
    OwnerFetchArgs a;
    memset(&a, 0, sizeof(a));
    int gl = 0;
    const int st = fx_fill_tables(tables_host, n_tables, a.t, &gl, "fx_owner_fetch_rows", catchup != 0);
    if (st != FX_OK) return st;
    int used = 0;
    for (int t = 0; t < n_tables; ++t) {
        FX_CHECK_ARG(off_host[t] >= 0 && off_host[t] + a.t[t].D <= ld && off_host[t] % a.t[t].vec == 0 &&
                         ld % a.t[t].vec == 0,
                     "fx_owner_fetch_rows: table %d does not fit / align in the block", t);
        a.off[t] = off_host[t];
        if (off_host[t] + a.t[t].D > used) used = off_host[t] + a.t[t].D;
    }
    a.uniq_row = uniq_row; a.seg_start = seg_start; a.sorted_pos = sorted_pos; a.n_unique = n_unique;
    a.send = send; a.ld = ld; a.n_total = n_total; a.scal = scal; a.n_tables = n_tables;
    a.group_log2 = gl; a.upto_offset = upto_offset; a.catchup = catchup ? 1 : 0; a.used_w = used;
    a.zero_row = zero_row; a.zero_w = zero_w;
    a.quad = !catchup || !fx_catchup_quad_on || gl != 2 || a.t[0].vec != 4 || a.t[0].D != 16 ? 0
             : (n_tables == 2 && a.t[1].vec == 1 && a.t[1].D == 1) ? 1 : n_tables == 1 ? 2 : 0;
    int64_t blocks = fx_ceil_div(n_total, 256 >> gl);
    if (blocks > 256 * 64) blocks = 256 * 64;
    hipLaunchKernelGGL(k_owner_fetch_rows, dim3((unsigned)blocks), dim3(256), 0, fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}

// ---------------------------------------------------------------------------------------------
// fx_pack_columns_multi: like fx_pack_columns, but every column names its own destination matrix,
// so the id block, the numeric block and the label of one batch are cast in ONE launch.
// ---------------------------------------------------------------------------------------------
#define FX_PACKM_MAX_COLS 96
struct PackMultiArgs {
    const void* col[FX_PACKM_MAX_COLS];
    void* out[FX_PACKM_MAX_COLS];          // first element of the column's destination
    int32_t ld[FX_PACKM_MAX_COLS];
    int16_t width[FX_PACKM_MAX_COLS];
    int8_t dtype[FX_PACKM_MAX_COLS];
    int8_t out_dtype[FX_PACKM_MAX_COLS];
    int64_t B;
};

__global__ __launch_bounds__(256) void k_pack_columns_multi(PackMultiArgs a) {
    const int c = blockIdx.y;
    const int64_t w = a.width[c];
    const int64_t n = a.B * w;
    const int dt = a.dtype[c];
    const void* src = a.col[c];
    const int64_t ld = a.ld[c];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / w, k = i - b * w;
        if (a.out_dtype[c] == FX_I32) {
            int32_t v;
            if (dt == FX_F64) v = (int32_t) reinterpret_cast<const double*>(src)[i];
            else if (dt == FX_I64) v = (int32_t) reinterpret_cast<const int64_t*>(src)[i];
            else if (dt == FX_F32) v = (int32_t) reinterpret_cast<const float*>(src)[i];
            else v = reinterpret_cast<const int32_t*>(src)[i];
            reinterpret_cast<int32_t*>(a.out[c])[b * ld + k] = v;
        } else {
            float v;
            if (dt == FX_F64) v = (float) reinterpret_cast<const double*>(src)[i];
            else if (dt == FX_I64) v = (float) reinterpret_cast<const int64_t*>(src)[i];
            else if (dt == FX_F32) v = reinterpret_cast<const float*>(src)[i];
            else v = (float) reinterpret_cast<const int32_t*>(src)[i];
            reinterpret_cast<float*>(a.out[c])[b * ld + k] = v;
        }
    }
}

extern "C" int fx_pack_columns_multi(const void* const* cols_host, const int32_t* dtypes_host,
                                     const int32_t* widths_host, void* const* outs_host,
                                     const int32_t* out_dtypes_host, const int64_t* out_lds_host,
                                     int32_t ncols, int64_t B, fx_stream_t stream) {
    FX_CHECK_ARG(ncols >= 0 && ncols <= FX_PACKM_MAX_COLS,
                 "fx_pack_columns_multi: ncols=%d not in [0,%d]", ncols, FX_PACKM_MAX_COLS);
    FX_CHECK_ARG(B >= 0, "fx_pack_columns_multi: B < 0");
    if (ncols == 0 || B == 0) return FX_OK;
    FX_CHECK_ARG(cols_host && dtypes_host && widths_host && outs_host && out_dtypes_host &&
                     out_lds_host, "fx_pack_columns_multi: null pointer");
    PackMultiArgs a;
    memset(&a, 0, sizeof(a));
    int64_t maxw = 1;
    for (int c = 0; c < ncols; ++c) {
        FX_CHECK_ARG(cols_host[c] && outs_host[c], "fx_pack_columns_multi: column %d is null", c);
        FX_CHECK_ARG(dtypes_host[c] >= FX_F32 && dtypes_host[c] <= FX_I64,
                     "fx_pack_columns_multi: bad dtype %d for column %d", dtypes_host[c], c);
        FX_CHECK_ARG(out_dtypes_host[c] == FX_I32 || out_dtypes_host[c] == FX_F32,
                     "fx_pack_columns_multi: out dtype of column %d must be FX_I32 or FX_F32", c);
        FX_CHECK_ARG(widths_host[c] >= 1 && widths_host[c] <= 32767 &&
                         out_lds_host[c] >= widths_host[c] && out_lds_host[c] < (1ll << 31),
                     "fx_pack_columns_multi: bad width / ld of column %d", c);
        a.col[c] = cols_host[c];
        a.out[c] = outs_host[c];
        a.ld[c] = (int32_t)out_lds_host[c];
        a.width[c] = (int16_t)widths_host[c];
        a.dtype[c] = (int8_t)dtypes_host[c];
        a.out_dtype[c] = (int8_t)out_dtypes_host[c];
        if (widths_host[c] > maxw) maxw = widths_host[c];
    }
    a.B = B;
    int64_t gx = fx_ceil_div(B * maxw, 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(k_pack_columns_multi, dim3((unsigned)gx, (unsigned)ncols), dim3(256), 0,
                       fx_hip_stream(stream), a);
    FX_CHECK_LAUNCH();
    return FX_OK;
}
