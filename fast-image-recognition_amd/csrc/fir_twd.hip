// fir_twd.hip -- the three-way-decision classifiers of qt_cpp/ImageTesting.cpp:74-288 on gfx950.
//
// The N x d' distance scans are the library's range-distance kernel (fir_range_distances_dev:
// one lane per gallery row, reference arithmetic order). What is added here is the reference's
// decision logic, one 256-thread workgroup per query, as order-exact parallel forms of its
// sequential loops:
//   * the running first-minimum (strict '<' from 100000) is a prefix scan of (distance, row)
//     pairs with combine(earlier, later) = later.d < earlier.d ? later : earlier;
//   * `secondBestDist` (ImageTesting.cpp:123-125: "the best so far, at the moment the best moved to
//     a row of another class") is the exclusive-prefix minimum seen by the LAST record-setting row
//     whose class differs from its predecessor record's class;
//   * per-class posteriors max exp(-100 d) (:118-122) are 64-bit LDS atomic max on the bit pattern
//     (non-negative doubles order like unsigned integers).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kBatch = 64;         // queries per internal batch at most: all their scans and decision workgroups are queued
                                   // before ONE synchronisation (a range-distance pass serves 8 of them)
constexpr int kLastFeature = 256;  // ImageTesting.cpp:169-171, 226-228

thread_local char g_twd_err[512];
int twd_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_twd_err, sizeof(g_twd_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_twd_err);
    return code;
}
#define TWD_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return twd_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,        \
                                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct DI {
    double d;
    int i;
};
__device__ __forceinline__ DI later_wins_if_smaller(const DI earlier, const DI later) { return later.d < earlier.d ? later : earlier; }
__device__ __forceinline__ DI shfl_up_di(const DI v, int off) {
    DI o;
    o.d = __shfl_up(v.d, off, 64);
    o.i = __shfl_up(v.i, off, 64);
    return o;
}
// lexicographic (value, row) minimum over the block; every thread gets the result
__device__ DI block_argmin(DI v, DI* red /* [kBlock/64] in LDS */) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        DI o;
        o.d = __shfl_xor(v.d, off, 64);
        o.i = __shfl_xor(v.i, off, 64);
        if (o.d < v.d || (o.d == v.d && (unsigned)o.i < (unsigned)v.i)) v = o;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    DI r = red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w)
        if (red[w].d < r.d || (red[w].d == r.d && (unsigned)red[w].i < (unsigned)r.i)) r = red[w];
    return r;
}

// ---- ConventionalTWDClassifier, first stage + reliability test (ImageTesting.cpp:108-164) ----
// dist1[q][n]: distances over [0, reduced_features_count). Dynamic LDS: num_classes doubles.
__global__ void __launch_bounds__(kBlock) k_twd_conv_stage1(const float* __restrict__ dist1, const int32_t* __restrict__ cls, int n,
                                                             int num_classes, int type, double threshold, int32_t* __restrict__ class_out,
                                                             int32_t* __restrict__ unreliable_out) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long probabs[];   // bit patterns of non-negative doubles
    __shared__ DI wave_tot[kBlock / 64];
    __shared__ DI red[kBlock / 64];
    __shared__ DI carry_s;
    const int q = blockIdx.x;
    const float* d1 = dist1 + (size_t)q * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int c = threadIdx.x; c < num_classes; c += kBlock) probabs[c] = 0ull;     // vector<double> probabs(num_of_classes) = 0
    if (threadIdx.x == 0) { carry_s.d = 100000.0; carry_s.i = -1; }                // bestDist = 100000, bestInd = -1 (:110-111)
    __syncthreads();
    // The rows are taken kSpan = 256 x kPer at a time: loaded coalesced into LDS, then thread t owns the CONTIGUOUS rows
    // [t*kPer, (t+1)*kPer) of the span, so the reference's scan order holds inside a thread, across the threads and
    // across spans, with one block-wide scan per span (not per 256 rows). kPer is odd: conflict-free LDS reads.
    constexpr int kPer = 15, kSpan = kBlock * kPer;
    __shared__ float sd[kSpan];
    __shared__ int32_t sc[kSpan];
    int last_change_row = -1;          // last record row whose class differs from the previous record's class
    double second_at_change = 0.0;     // what secondBestDist was set to at that row (:124-125)
    for (int base = 0; base < n; base += kSpan) {
        const int live = min(kSpan, n - base);
        for (int i = threadIdx.x; i < live; i += kBlock) { sd[i] = d1[base + i]; sc[i] = cls[base + i]; }
        __syncthreads();
        const int r0 = min(threadIdx.x * kPer, live), r1 = min(r0 + kPer, live);
        // pass 1: the segment's first minimum; class posteriors (order independent: a maximum), one LDS atomic per run of
        // equal labels (galleries are class-major, ImageTesting.cpp:446)
        DI own;
        own.d = __builtin_huge_val();
        own.i = -1;
        int run_class = -1;
        double run_max = 0.0;
        for (int r = r0; r < r1; ++r) {
            const double d = (double)sd[r];                                         // distances[j] (double) (:117)
            if (d < own.d) { own.d = d; own.i = base + r; }
            if (type == 0) {
                const double probab = exp(-d * 100);                                // DIST_WEIGHT = 100 (:113,119)
                const int cl = sc[r];
                if (cl != run_class) {
                    if (run_class >= 0 && run_class < num_classes) atomicMax(&probabs[run_class], (unsigned long long)__double_as_longlong(run_max));
                    run_class = cl;
                    run_max = probab;
                } else if (run_max < probab) run_max = probab;                      // :120-121
            }
        }
        if (type == 0 && run_class >= 0 && run_class < num_classes) atomicMax(&probabs[run_class], (unsigned long long)__double_as_longlong(run_max));
        // exclusive scan of the segment minima, seeded with what the earlier spans left behind
        DI inc = own;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const DI o = shfl_up_di(inc, off);
            if (lane >= off) inc = later_wins_if_smaller(o, inc);
        }
        if (lane == 63) wave_tot[wave] = inc;
        __syncthreads();
        DI pre = carry_s;
        for (int w = 0; w < wave; ++w) pre = later_wins_if_smaller(pre, wave_tot[w]);
        DI excl = shfl_up_di(inc, 1);
        excl = lane == 0 ? pre : later_wins_if_smaller(pre, excl);
        // pass 2: the reference's loop over this thread's rows, entered with the state the earlier rows left behind
        if (own.i >= 0 && own.d < excl.d) {      // otherwise no row of this segment sets a record
            DI cur = excl;
            int cur_class = cur.i >= 0 ? cls[cur.i] : -1;
            for (int r = r0; r < r1; ++r) {
                const double d = (double)sd[r];
                if (d < cur.d) {                                                     // a new best (:123)
                    const int cl = sc[r];
                    if (cur.i != -1 && cur_class != cl) { last_change_row = base + r; second_at_change = cur.d; }   // :124-125
                    cur.d = d; cur.i = base + r; cur_class = cl;
                }
            }
        }
        __syncthreads();
        if (threadIdx.x == kBlock - 1) carry_s = later_wins_if_smaller(pre, inc);
        __syncthreads();
    }
    const DI best = carry_s;
    DI lc;
    lc.d = -(double)last_change_row;     // arg-MAX of the row through the arg-min helper
    lc.i = last_change_row;
    const DI lcw = block_argmin(lc, red);
    // the thread that owns the winning row publishes its value
    __shared__ double second_s;
    if (threadIdx.x == 0) second_s = 100000.0;                                      // secondBestDist = 100000 (:111)
    __syncthreads();
    if (lcw.i >= 0 && last_change_row == lcw.i) second_s = second_at_change;
    __syncthreads();
    const double secondBest = second_s;

    bool reliable = false;
    if (best.i >= 0) {
        if (type == 0) {
            // sum of the 5 largest class posteriors (:141-146), taken in descending order
            double sum = 0.0;
            for (int r = 0; r < 5; ++r) {
                DI m;
                m.d = __builtin_huge_val();
                m.i = -1;
                for (int c = threadIdx.x; c < num_classes; c += kBlock) {
                    const double v = __longlong_as_double((long long)probabs[c]);
                    if (probabs[c] != ~0ull && (-v < m.d)) { m.d = -v; m.i = c; }
                }
                const DI w = block_argmin(m, red);
                sum += -w.d;
                __syncthreads();
                if (threadIdx.x == 0 && w.i >= 0) probabs[w.i] = ~0ull;             // taken
                __syncthreads();
            }
            const double max_probab = exp(-best.d * 100) / sum;                     // :130,147
            reliable = max_probab > threshold;                                      // :148
        } else if (type == 1) {
            reliable = (secondBest - best.d) > threshold;                           // :158
        } else {
            reliable = (best.d / secondBest) < threshold;                           // :161
        }
    }
    if (threadIdx.x == 0) {
        unreliable_out[q] = reliable ? 0 : 1;
        class_out[q] = best.i >= 0 ? cls[best.i] : -1;                              // overwritten by the second stage when unreliable
    }
}

// ---- second stage (ImageTesting.cpp:165-180) for the unreliable queries ----
// One workgroup per query of the batch; the reliable ones (unreliable[slot] == 0) return at once, so the host does not
// have to look at the first stage's verdicts before queueing this. dist1 / dist2: [slot][n] over [0, reduced) / [reduced, 256).
__global__ void __launch_bounds__(kBlock) k_twd_conv_stage2(const float* __restrict__ dist1, const float* __restrict__ dist2,
                                                             const int32_t* __restrict__ unreliable, const int32_t* __restrict__ cls, int n,
                                                             int reduced, int32_t* __restrict__ class_out) {
    __shared__ DI red[kBlock / 64];
    const int slot = blockIdx.x;
    if (!unreliable[slot]) return;
    const float* d1 = dist1 + (size_t)slot * n;
    const float* d2 = dist2 + (size_t)slot * n;
    DI m;
    m.d = 100000.0;      // bestDist = 100000 (:168); only strictly smaller rows qualify
    m.i = -1;
    for (int row = threadIdx.x; row < n; row += kBlock) {
        const float tail = d2[row] * (float)(kLastFeature - reduced);               // float * int -> float (:174)
        const double v = ((double)d1[row] * reduced + tail) / kLastFeature;         // :173-174
        if (v < m.d || (v == m.d && m.i >= 0 && row < m.i)) { m.d = v; m.i = row; }
    }
    const DI w = block_argmin(m, red);
    if (threadIdx.x == 0) class_out[slot] = w.i >= 0 ? cls[w.i] : -1;
}

// ---- ProposedTWDClassifier (ImageTesting.cpp:207-288, CHECK_ALL_INSTANCES) ----
// cd[c][slot][n]: distances of chunk c = features [c*reduced, (c+1)*reduced); acc[slot][n] doubles and
// alive[slot][n] bytes are workspace (any content on entry).
__global__ void __launch_bounds__(kBlock) k_twd_proposed(const float* __restrict__ cd, int nq, int nchunks, double* __restrict__ acc,
                                                          uint8_t* __restrict__ alive, const int32_t* __restrict__ cls, int n,
                                                          double threshold /* 1/th */, int32_t* __restrict__ class_out,
                                                          int32_t* __restrict__ unreliable_out, int32_t* __restrict__ chunks_out) {
    __shared__ DI red[kBlock / 64];
    __shared__ int cnt_s;
    const int q = blockIdx.x;
    double* a = acc + (size_t)q * n;
    uint8_t* live = alive + (size_t)q * n;
    for (int row = threadIdx.x; row < n; row += kBlock) { a[row] = 0.0; live[row] = 1; }   // :209,216
    int bestInd = -1, unreliable = 0, used = 0;
    for (int c = 0; c < nchunks; ++c) {
        ++used;
        const float* dc = cd + ((size_t)c * nq + q) * n;
        DI m;
        m.d = 100000.0;                                                             // bestDist = 100000 per chunk (:230)
        m.i = -1;
        for (int row = threadIdx.x; row < n; row += kBlock) {
            if (!live[row]) continue;                                               // :236-241
            const double v = a[row] + (double)dc[row];                              // distances[j] += ... (:250)
            a[row] = v;
            if (v < m.d) { m.d = v; m.i = row; }                                    // rows ascend per thread: strict '<' keeps the first
        }
        const DI w = block_argmin(m, red);
        if (w.i >= 0) bestInd = w.i;                                                // :255-258
        if (bestInd < 0) break;
        const double dist_threshold = w.d * threshold;                              // :263 (bestDist stays 100000 when nothing qualified)
        const int bestClass = cls[bestInd];
        if (threadIdx.x == 0) cnt_s = 0;
        __syncthreads();
        int others = 0;
        for (int row = threadIdx.x; row < n; row += kBlock) {
            if (!live[row]) continue;
            if (a[row] > dist_threshold) live[row] = 0;                             // :268-269
            else if (cls[row] != bestClass) ++others;                               // :270-271
        }
        atomicAdd(&cnt_s, others);
        __syncthreads();
        const int num_of_variants = 1 + cnt_s;                                      // :265
        __syncthreads();
        if (num_of_variants == 1) break;                                            // :285
        if (c == 0) ++unreliable;                                                   // :287-288
    }
    if (threadIdx.x == 0) {
        class_out[q] = bestInd >= 0 ? cls[bestInd] : -1;
        unreliable_out[q] = unreliable;
        chunks_out[q] = used;
    }
}

// Queries per internal batch: as many as keep the per-batch distance tables under `budget` bytes (a multiple of 8, <= kBatch).
int batch_for(int64_t n, size_t bytes_per_query_row, size_t budget = (size_t)512 << 20) {
    const size_t per_query = (size_t)std::max<int64_t>(n, 1) * bytes_per_query_row;
    const size_t fit = budget / std::max<size_t>(per_query, 1);
    return (int)std::max<size_t>(8, std::min<size_t>(kBatch, fit / 8 * 8));
}

// A slot of the gallery handle's scratch pool (fir_gallery_scratch_): grown on demand, reused by every later call.
struct Slot {
    void* p = nullptr;
    template <typename T> T* as() { return (T*)p; }
};
#define TWD_SLOT(var, slot, bytes)                                                       \
    Slot var;                                                                            \
    if ((rc = fir_gallery_scratch_(g, slot, std::max<size_t>(bytes, 16), &var.p))) return rc

int check_common(fir_gallery* g, const float* queries, int32_t qb, int32_t reduced, fir_gallery_view* v) {
    if (!g || (qb > 0 && !queries)) return twd_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return twd_fail(FIR_ERR_ARG, "qb < 0");
    if (fir_gallery_view_(g, v) != FIR_OK) return twd_fail(FIR_ERR_ARG, "bad gallery");
    if (!v->cls) return twd_fail(FIR_ERR_STATE, "gallery was created without class labels");
    if (v->d < kLastFeature) return twd_fail(FIR_ERR_ARG, "the TWD classifiers use features [0,%d); the gallery has %d", kLastFeature, v->d);
    if (reduced <= 0 || reduced >= kLastFeature) return twd_fail(FIR_ERR_ARG, "reduced_features_count=%d outside (0,%d)", reduced, kLastFeature);
    if (v->n >= (int64_t)1 << 30) return twd_fail(FIR_ERR_ARG, "gallery too large for the TWD drivers");
    return FIR_OK;
}

}  // namespace

extern "C" {

int fir_twd_conventional(fir_gallery* g, const float* queries, int32_t qb, int32_t num_classes, int32_t type, double threshold,
                         int32_t reduced_features_count, int32_t* class_out, int32_t* unreliable_out) {
    fir_gallery_view v;
    int rc = check_common(g, queries, qb, reduced_features_count, &v);
    if (rc) return rc;
    if (!class_out) return twd_fail(FIR_ERR_ARG, "class_out is NULL");
    if (type < 0 || type > 2) return twd_fail(FIR_ERR_ARG, "type %d outside [0,2]", type);
    if (num_classes < 5 || (size_t)num_classes * 8 > 60 * 1024)
        return twd_fail(FIR_ERR_ARG, "num_classes=%d outside [5, 7680] (top-5 posteriors, ImageTesting.cpp:141; LDS table)", num_classes);
    if (qb == 0) return FIR_OK;
    TWD_HIP(hipSetDevice(v.device));
    const int n = (int)v.n;
    const int batch = std::min(batch_for(n, 8), std::max(8, (qb + 7) / 8 * 8));
    TWD_SLOT(dq, 0, (size_t)batch * v.d * 4);
    TWD_SLOT(d1, 1, (size_t)batch * std::max(n, 1) * 4);
    TWD_SLOT(d2, 2, (size_t)batch * std::max(n, 1) * 4);
    TWD_SLOT(dres, 3, (size_t)2 * kBatch * 4);                 // class[kBatch], unreliable[kBatch]
    int32_t* dcls = dres.as<int32_t>();
    int32_t* dunrel = dcls + kBatch;
    for (int q0 = 0; q0 < qb; q0 += batch) {
        const int nq = std::min(batch, qb - q0);
        int32_t h_res[2 * kBatch];
        if (n == 0) {
            for (int i = 0; i < nq; ++i) { h_res[i] = -1; h_res[kBatch + i] = 1; }
        } else {
            // both stages are queued back to back -- the second one decides on the device which queries it concerns -- and
            // the verdicts come back with ONE copy and ONE synchronisation per batch
            TWD_HIP(hipMemcpyAsync(dq.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
            if ((rc = fir_range_distances_dev(g, dq.as<float>(), nq, 0, reduced_features_count, d1.as<float>(), v.stream))) return rc;
            hipLaunchKernelGGL(k_twd_conv_stage1, dim3(nq), dim3(kBlock), (size_t)num_classes * 8, v.stream, d1.as<float>(), v.cls, n,
                               num_classes, type, threshold, dcls, dunrel);
            TWD_HIP(hipGetLastError());
            if ((rc = fir_range_distances_dev(g, dq.as<float>(), nq, reduced_features_count, kLastFeature, d2.as<float>(), v.stream))) return rc;
            hipLaunchKernelGGL(k_twd_conv_stage2, dim3(nq), dim3(kBlock), 0, v.stream, d1.as<float>(), d2.as<float>(), dunrel, v.cls, n,
                               reduced_features_count, dcls);
            TWD_HIP(hipGetLastError());
            TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
            TWD_HIP(hipStreamSynchronize(v.stream));
        }
        for (int i = 0; i < nq; ++i) {
            class_out[q0 + i] = h_res[i];
            if (unreliable_out) unreliable_out[q0 + i] = h_res[kBatch + i];
        }
    }
    return FIR_OK;
}

int fir_twd_proposed(fir_gallery* g, const float* queries, int32_t qb, int32_t reduced_features_count, double threshold,
                     int32_t* class_out, int32_t* unreliable_out, int32_t* chunks_out) {
    fir_gallery_view v;
    int rc = check_common(g, queries, qb, reduced_features_count, &v);
    if (rc) return rc;
    if (!class_out) return twd_fail(FIR_ERR_ARG, "class_out is NULL");
    if (!(threshold > 0)) return twd_fail(FIR_ERR_ARG, "threshold must be > 0");
    if (qb == 0) return FIR_OK;
    TWD_HIP(hipSetDevice(v.device));
    const int n = (int)v.n;
    // chunks cover [0,256) in steps of reduced_features_count; the reference reads past 256 when the step does not
    // divide it (ImageTesting.cpp:229,250) -- only steps that divide 256 are accepted here
    if (kLastFeature % reduced_features_count != 0)
        return twd_fail(FIR_ERR_ARG, "reduced_features_count=%d must divide %d", reduced_features_count, kLastFeature);
    const int nchunks = kLastFeature / reduced_features_count;
    const int batch = std::min(batch_for(n, (size_t)nchunks * 4 + 9, (size_t)1 << 30), std::max(8, (qb + 7) / 8 * 8));
    TWD_SLOT(dq, 0, (size_t)batch * v.d * 4);
    TWD_SLOT(cd, 4, (size_t)nchunks * batch * std::max(n, 1) * 4);
    TWD_SLOT(acc, 5, (size_t)batch * std::max(n, 1) * 8);
    TWD_SLOT(alive, 6, (size_t)batch * std::max(n, 1));
    TWD_SLOT(dres, 3, (size_t)3 * kBatch * 4);                 // class, unreliable, chunks
    int32_t* dcls = dres.as<int32_t>();
    int32_t* dunrel = dcls + kBatch;
    int32_t* dchunks = dcls + 2 * kBatch;
    for (int q0 = 0; q0 < qb; q0 += batch) {
        const int nq = std::min(batch, qb - q0);
        int32_t h_res[3 * kBatch];
        int32_t* h_cls = h_res;
        int32_t* h_unrel = h_res + kBatch;
        int32_t* h_chunks = h_res + 2 * kBatch;
        if (n == 0) {
            for (int i = 0; i < nq; ++i) { h_cls[i] = -1; h_unrel[i] = 0; h_chunks[i] = 0; }
        } else {
            TWD_HIP(hipMemcpyAsync(dq.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
            // all chunk distances cd[c][slot][n] from ONE pass over features [0, 256)
            if ((rc = fir_subrange_distances_dev_(g, dq.as<float>(), nq, 0, kLastFeature, reduced_features_count, cd.as<float>(), v.stream))) return rc;
            hipLaunchKernelGGL(k_twd_proposed, dim3(nq), dim3(kBlock), 0, v.stream, cd.as<float>(), nq, nchunks, acc.as<double>(),
                               alive.as<uint8_t>(), v.cls, n, 1.0 / threshold, dcls, dunrel, dchunks);
            TWD_HIP(hipGetLastError());
            TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
            TWD_HIP(hipStreamSynchronize(v.stream));
        }
        for (int i = 0; i < nq; ++i) {
            class_out[q0 + i] = h_cls[i];
            if (unreliable_out) unreliable_out[q0 + i] = h_unrel[i];
            if (chunks_out) chunks_out[q0 + i] = h_chunks[i];
        }
    }
    return FIR_OK;
}

}  // extern "C"
