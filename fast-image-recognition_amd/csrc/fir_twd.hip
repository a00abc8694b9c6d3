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
#include <cstring>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"
#include "fir_common.h"

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
//
// MODE kS1Single: one workgroup per query does everything (galleries up to a few thousand rows: one launch).
// Larger galleries split the rows of a query over gridDim.x workgroups, in three launches:
//   kS1Part     each workgroup: first minimum of its row segment -> part[q][b]; class posteriors into gprob[q][C]
//               (a maximum: order independent; 64-bit atomic max on the bit pattern of a non-negative double)
//   kS1Records  each workgroup: the state the earlier segments leave behind = fold of part[q][0..b), then the
//               reference's record walk over its own rows -> chg[q][b] = last class-changing record and the
//               secondBestDist it set
//   kS1Final    one workgroup per query: best row = fold of all segments, secondBestDist = the LAST segment's change,
//               top-5 posteriors, the reliability test, the outputs
// The second stage (:165-180) needs no launches of its own: both partial distances come from ONE gallery pass
// (fir_split_distances_dev_), kS1Part also leaves each segment's second-stage minimum behind, and the deciding workgroup
// (kS1Single / kS1Final) takes it when the reliability test fails -- and, on a one-query call, writes the verdict and the
// ticket to pinned host memory itself. 100 000 x 512, one query: 13 launches / 99 us -> 6 / 81 us; 3 030 x 1536: 51 -> 45 us.
enum { kS1Single = 0, kS1Part = 1, kS1Records = 2, kS1Final = 3 };
struct S1Chg {
    int row;
    int pad;
    double second;
};
constexpr int kPer = 15, kSpan = kBlock * kPer;     // rows per block-wide scan; kPer odd: conflict-free LDS reads

template <int MODE>
__global__ void __launch_bounds__(kBlock) k_twd_conv_stage1(const float* __restrict__ dist1, const int32_t* __restrict__ cls, int n,
                                                             int num_classes, int type, double threshold, int32_t* __restrict__ class_out,
                                                             int32_t* __restrict__ unreliable_out, DI* __restrict__ part,
                                                             S1Chg* __restrict__ chg, unsigned long long* __restrict__ gprob, int seg_rows,
                                                             const float* __restrict__ dist2, int reduced, DI* __restrict__ part2,
                                                             int32_t* __restrict__ host_res, int host_stride, uint64_t* __restrict__ host_ticket,
                                                             uint64_t ticket) {
    // dist2[q][n]: distances over [reduced, 256), already there -- the second stage (ImageTesting.cpp:165-180) rides along:
    // kS1Part leaves each segment's second-stage minimum in part2 (for every query: whether it is needed is only known at
    // the end), kS1Final / kS1Single take it when the reliability test fails.
    // host_res (one-query calls): class and verdict also go to pinned host memory, then `ticket` to host_ticket.
    extern __shared__ __attribute__((aligned(16))) unsigned long long probabs[];   // bit patterns of non-negative doubles
    __shared__ DI wave_tot[kBlock / 64];
    __shared__ DI red[kBlock / 64];
    __shared__ DI carry_s;
    __shared__ float sd[kSpan];
    __shared__ int32_t sc[kSpan];
    const int q = MODE == kS1Single || MODE == kS1Final ? blockIdx.x : blockIdx.y;
    const int nseg = MODE == kS1Single ? 1 : (n + seg_rows - 1) / seg_rows;
    const int b = MODE == kS1Part || MODE == kS1Records ? blockIdx.x : 0;
    const float* d1 = dist1 + (size_t)q * n;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool with_prob = type == 0 && (MODE == kS1Single || MODE == kS1Part || MODE == kS1Final);
    if (with_prob)
        for (int c = threadIdx.x; c < num_classes; c += kBlock)
            probabs[c] = MODE == kS1Final ? gprob[(size_t)q * num_classes + c] : 0ull;     // vector<double> probabs(num_of_classes) = 0
    // rows of this workgroup and the state the rows before them leave behind
    const int row_begin = MODE == kS1Single ? 0 : b * seg_rows;
    const int row_end = MODE == kS1Single ? n : min(n, row_begin + seg_rows);
    if (threadIdx.x == 0) {
        DI in;
        in.d = 100000.0;                                                            // bestDist = 100000, bestInd = -1 (:110-111)
        in.i = -1;
        if (MODE == kS1Records)
            for (int k = 0; k < b; ++k) in = later_wins_if_smaller(in, part[(size_t)q * nseg + k]);
        if (MODE == kS1Final)
            for (int k = 0; k < nseg; ++k) in = later_wins_if_smaller(in, part[(size_t)q * nseg + k]);
        carry_s = in;
    }
    __syncthreads();
    DI m2;                             // second stage: bestDist = 100000 (:168), only strictly smaller rows qualify
    m2.d = 100000.0;
    m2.i = -1;
    int last_change_row = -1;          // last record row whose class differs from the previous record's class
    double second_at_change = 0.0;     // what secondBestDist was set to at that row (:124-125)
    if (MODE != kS1Final) {
        // The rows are taken kSpan at a time: loaded coalesced into LDS, then thread t owns the CONTIGUOUS rows
        // [t*kPer, (t+1)*kPer) of the span, so the reference's scan order holds inside a thread, across the threads and
        // across spans, with one block-wide scan per span (not per 256 rows).
        for (int base = row_begin; base < row_end; base += kSpan) {
            const int live = min(kSpan, row_end - base);
            for (int i = threadIdx.x; i < live; i += kBlock) {
                const float dv = d1[base + i];
                sd[i] = dv;
                sc[i] = cls[base + i];
                if (MODE == kS1Part) {                                              // :173-174 for row base + i (rows ascend per thread)
                    const float tail = dist2[(size_t)q * n + base + i] * (float)(kLastFeature - reduced);
                    const double v = ((double)dv * reduced + tail) / kLastFeature;
                    if (v < m2.d) { m2.d = v; m2.i = base + i; }
                }
            }
            __syncthreads();
            const int r0 = min(threadIdx.x * kPer, live), r1 = min(r0 + kPer, live);
            // pass 1: the segment's first minimum; class posteriors, one LDS atomic per run of equal labels (galleries
            // are class-major, ImageTesting.cpp:446)
            DI own;
            own.d = __builtin_huge_val();
            own.i = -1;
            int run_class = -1;
            double run_max = 0.0;
            for (int r = r0; r < r1; ++r) {
                const double d = (double)sd[r];                                     // distances[j] (double) (:117)
                if (d < own.d) { own.d = d; own.i = base + r; }
                if (with_prob) {
                    const double probab = exp(-d * 100);                            // DIST_WEIGHT = 100 (:113,119)
                    const int cl = sc[r];
                    if (cl != run_class) {
                        if (run_class >= 0 && run_class < num_classes) atomicMax(&probabs[run_class], (unsigned long long)__double_as_longlong(run_max));
                        run_class = cl;
                        run_max = probab;
                    } else if (run_max < probab) run_max = probab;                  // :120-121
                }
            }
            if (with_prob && run_class >= 0 && run_class < num_classes) atomicMax(&probabs[run_class], (unsigned long long)__double_as_longlong(run_max));
            // exclusive scan of the thread minima, seeded with what the earlier rows left behind
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
            if (MODE != kS1Part && own.i >= 0 && own.d < excl.d) {      // otherwise no row of this thread sets a record
                DI cur = excl;
                int cur_class = cur.i >= 0 ? cls[cur.i] : -1;
                for (int r = r0; r < r1; ++r) {
                    const double d = (double)sd[r];
                    if (d < cur.d) {                                                 // a new best (:123)
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
    }
    if (MODE == kS1Part) {
        // this segment's own first minimum (the seed (100000, -1) wins over rows that are not below it, like the reference's
        // initial state does) and its share of the class posteriors
        if (threadIdx.x == 0) part[(size_t)q * nseg + b] = carry_s;
        const DI w2 = block_argmin(m2, red);
        if (threadIdx.x == 0) part2[(size_t)q * nseg + b] = w2;
        if (with_prob)
            for (int c = threadIdx.x; c < num_classes; c += kBlock)
                if (probabs[c]) atomicMax(&gprob[(size_t)q * num_classes + c], probabs[c]);
        return;
    }
    const DI best = carry_s;
    DI lc;
    lc.d = -(double)last_change_row;     // arg-MAX of the row through the arg-min helper
    lc.i = last_change_row;
    const DI lcw = block_argmin(lc, red);
    // the thread that owns the winning row publishes its value
    __shared__ double second_s;
    __shared__ int second_row_s;
    if (threadIdx.x == 0) { second_s = 100000.0; second_row_s = -1; }               // secondBestDist = 100000 (:111)
    __syncthreads();
    if (lcw.i >= 0 && last_change_row == lcw.i) { second_s = second_at_change; second_row_s = lcw.i; }
    __syncthreads();
    if (MODE == kS1Records) {
        if (threadIdx.x == 0) { chg[(size_t)q * nseg + b].row = second_row_s; chg[(size_t)q * nseg + b].second = second_s; }
        return;
    }
    if (MODE == kS1Final && threadIdx.x == 0)
        for (int k = nseg - 1; k >= 0; --k)
            if (chg[(size_t)q * nseg + k].row >= 0) { second_s = chg[(size_t)q * nseg + k].second; break; }
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
    int cls_final = best.i >= 0 ? cls[best.i] : -1;
    if (!reliable) {                                                                // block-uniform
        // second stage (:165-180): the row with the smallest distance over [0, 256), rebuilt from the two partial distances
        DI w;
        if (MODE == kS1Single) {
            const float* d2 = dist2 + (size_t)q * n;
            DI m;
            m.d = 100000.0;
            m.i = -1;
            for (int row = threadIdx.x; row < n; row += kBlock) {
                const float tail = d2[row] * (float)(kLastFeature - reduced);       // float * int -> float (:174)
                const double v = ((double)d1[row] * reduced + tail) / kLastFeature; // :173-174
                if (v < m.d) { m.d = v; m.i = row; }
            }
            w = block_argmin(m, red);
        } else {
            w.d = 100000.0;
            w.i = -1;
            for (int k = 0; k < nseg; ++k) {             // segments in row order: an equal value later on does not replace the earlier row
                const DI p2 = part2[(size_t)q * nseg + k];
                if (p2.i >= 0 && p2.d < w.d) w = p2;
            }
        }
        cls_final = w.i >= 0 ? cls[w.i] : -1;
    }
    if (threadIdx.x == 0) {
        unreliable_out[q] = reliable ? 0 : 1;
        class_out[q] = cls_final;
        if (host_res) {
            host_res[q] = cls_final;
            host_res[host_stride + q] = reliable ? 0 : 1;
            __threadfence_system();
            __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- ProposedTWDClassifier (ImageTesting.cpp:207-288, CHECK_ALL_INSTANCES) ----
// cd[c][slot][n]: distances of chunk c = features [c*reduced, (c+1)*reduced); acc[slot][n] doubles and
// alive[slot][n] bytes are workspace (any content on entry).
__global__ void __launch_bounds__(kBlock) k_twd_proposed(const float* __restrict__ cd, int nq, int nchunks, double* __restrict__ acc,
                                                          uint8_t* __restrict__ alive, const int32_t* __restrict__ cls, int n,
                                                          double threshold /* 1/th */, int32_t* __restrict__ class_out,
                                                          int32_t* __restrict__ unreliable_out, int32_t* __restrict__ chunks_out,
                                                          int32_t* __restrict__ host_res, int host_stride, uint64_t* __restrict__ host_ticket,
                                                          uint64_t ticket) {
    // host_res (one-query calls): the three outputs also go to pinned host memory, then `ticket` to host_ticket
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
        const int cl = bestInd >= 0 ? cls[bestInd] : -1;
        class_out[q] = cl;
        unreliable_out[q] = unreliable;
        chunks_out[q] = used;
        if (host_res) {
            host_res[q] = cl;
            host_res[host_stride + q] = unreliable;
            host_res[2 * host_stride + q] = used;
            __threadfence_system();
            __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ---- the same classifier with the rows of a query split over gridDim.x workgroups (large galleries): k_twd_prop_chunk ----
struct PropState {
    int bestInd, unreliable, used, done;
};
__global__ void k_twd_prop_init(PropState* state, int nq) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < nq) state[q] = PropState{-1, 0, 0, 0};
}
__device__ __forceinline__ DI prop_fold(const DI* __restrict__ parts, int nseg) {
    DI w;
    w.d = 100000.0;                                                                 // bestDist = 100000 per chunk (:230)
    w.i = -1;
    for (int b = 0; b < nseg; ++b) {
        const DI p = parts[b];
        if (p.i >= 0 && p.d < w.d) w = p;                                           // segments in row order, strict '<'
    }
    return w;
}
// The reference's loop bookkeeping for one chunk j whose segment minima (fold w) and surviving-other-class count are complete.
__device__ __forceinline__ void prop_advance(PropState& st, int j, const DI w, int others) {
    if (st.done) return;
    ++st.used;
    if (w.i >= 0) st.bestInd = w.i;                                                 // :255-258
    if (st.bestInd < 0) st.done = 1;
    else if (1 + others == 1) st.done = 1;                                          // num_of_variants == 1 (:265,285)
    else if (j == 0) ++st.unreliable;                                               // :287-288
}
// Launch c of nchunks + 1 (c = 0 .. nchunks), gridDim.x segments per query (blockIdx.y):
//   * every workgroup brings the query's state up to chunk c - 2 (whose minima and counts are complete) from the state launch
//     c - 1 left in s_in, workgroup 0 leaves it in s_out for launch c + 1 (two buffers: the others still read s_in);
//   * a finished query's workgroups return; the others PRUNE chunk c - 1 in their segment (rows above bestDist / threshold
//     die, the survivors of another class than the best row's are counted into cnt[c - 1]) ...
//   * ... and, speculatively -- whether chunk c - 1 ended the loop is only known once every segment has counted -- take the
//     segment's first minimum of chunk c over the rows still alive (launch c + 1 drops it when the loop had ended).
// One launch per chunk instead of three.
__global__ void __launch_bounds__(kBlock) k_twd_prop_chunk(const float* __restrict__ cd, int nq, int c, int nchunks, double* __restrict__ acc,
                                                            uint8_t* __restrict__ alive, const int32_t* __restrict__ cls, int n, int seg_rows,
                                                            double threshold, const PropState* __restrict__ s_in, PropState* __restrict__ s_out,
                                                            DI* __restrict__ part, int* __restrict__ cnt) {
    __shared__ DI red[kBlock / 64];
    __shared__ DI best_s;
    __shared__ int others_s, go_s, best_ind_s;
    const int q = blockIdx.y, b = blockIdx.x, nseg = gridDim.x;
    double* a = acc + (size_t)q * n;
    uint8_t* live = alive + (size_t)q * n;
    const int row_begin = b * seg_rows, row_end = min(n, (b + 1) * seg_rows);
    if (c > 0) {
        if (threadIdx.x == 0) {
            PropState st = s_in[q];
            if (c >= 2) prop_advance(st, c - 2, prop_fold(part + ((size_t)(c - 2) * nq + q) * nseg, nseg), cnt[(size_t)(c - 2) * nq + q]);
            if (b == 0) s_out[q] = st;
            int go = !st.done;
            if (go) {
                const DI w = prop_fold(part + ((size_t)(c - 1) * nq + q) * nseg, nseg);
                const int bestInd = w.i >= 0 ? w.i : st.bestInd;                    // :255-258
                if (bestInd < 0) go = 0;                                            // the reference breaks before pruning
                best_s = w;
                best_ind_s = bestInd;
            }
            go_s = go;
            others_s = 0;
        }
        __syncthreads();
        if (!go_s) return;
        const double dist_threshold = best_s.d * threshold;                         // :263
        const int bestClass = cls[best_ind_s];
        int others = 0;
        for (int row = row_begin + threadIdx.x; row < row_end; row += kBlock) {
            if (c > 1 && !live[row]) continue;
            if (a[row] > dist_threshold) live[row] = 0;                             // :268-269
            else {
                if (c == 1) live[row] = 1;                                          // (the first prune writes every flag: the workspace comes uninitialised)
                if (cls[row] != bestClass) ++others;                                // :270-271
            }
        }
        atomicAdd(&others_s, others);
        __syncthreads();
        if (threadIdx.x == 0 && others_s) atomicAdd(&cnt[(size_t)(c - 1) * nq + q], others_s);
        if (c >= nchunks) return;
    }
    const float* dc = cd + ((size_t)c * nq + q) * n;
    DI m;
    m.d = 100000.0;                                                                 // bestDist = 100000 per chunk (:230)
    m.i = -1;
    for (int row = row_begin + threadIdx.x; row < row_end; row += kBlock) {         // the thread that pruned a row is the one that reads its flag
        if (c > 0 && !live[row]) continue;                                          // :236-241 (every row is alive in the first chunk)
        const double v = (c > 0 ? a[row] : 0.0) + (double)dc[row];                  // distances[j] += ... (:250)
        a[row] = v;
        if (v < m.d) { m.d = v; m.i = row; }
    }
    const DI w = block_argmin(m, red);
    if (threadIdx.x == 0) part[((size_t)c * nq + q) * nseg + b] = w;
}
// After launch nchunks: the last two chunks' bookkeeping and the outputs, one thread per query.
__global__ void k_twd_prop_finish(int nq, int nchunks, int nseg, const DI* __restrict__ part, const int* __restrict__ cnt, const int32_t* __restrict__ cls,
                                  const PropState* __restrict__ s_in, int32_t* __restrict__ class_out, int32_t* __restrict__ unreliable_out,
                                  int32_t* __restrict__ chunks_out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq) return;
    PropState st = s_in[q];                                                         // the state up to chunk nchunks - 2
    const int j = nchunks - 1;
    prop_advance(st, j, prop_fold(part + ((size_t)j * nq + q) * nseg, nseg), cnt[(size_t)j * nq + q]);
    class_out[q] = st.bestInd >= 0 ? cls[st.bestInd] : -1;
    unreliable_out[q] = st.unreliable;
    chunks_out[q] = st.used;
}

// ---- the same classifier as ONE launch per call (few queries): k_twd_prop_fused ----
// The forms above read n x 4 bytes per query and chunk that a scan wrote just before, and the split form is one launch per
// chunk. Here the chunk distances never leave the registers: blockIdx.y = query, the query's rows are dealt tile by tile
// (64 rows, lane = row: the tiled f32 gallery of fir_kernels.h) to the gridDim.x * 8 waves of its workgroups, T tiles per
// wave, and every lane keeps its rows' running double sums, class labels and alive bits for the whole call. Per chunk c the
// workgroups of a query meet ONCE:
//   before the meeting   each lane adds the chunk's distance (the reference's float loop and division, db_features.cpp:22-42)
//                        to its alive rows' sums; the workgroup's smallest sum goes into slot[c].vmin (atomic, memory side);
//   after it             everybody knows bestDist of chunk c, prunes its own rows against bestDist * threshold (:256-266),
//                        posts the class range of its survivors and -- the workgroup(s) that hold the minimum -- the first row
//                        at it with its class, and goes on to chunk c + 1 SPECULATIVELY: whether chunk c ended the loop
//                        (num_of_variants == 1: every survivor has the best row's class, :278) is known at the NEXT meeting, when
//                        slot[c]'s class range and best row are complete. A chunk's distances are only computed for tiles that
//                        still have a row alive -- what the reference's classifier is about.
// Everything the workgroups exchange goes through returning 64-bit atomics (the XCDs' L2s are not coherent with each other, and
// plain or sc1 loads of a word other workgroups update by atomics have been seen served from stale lines: fir_gemm_f16x.h),
// complemented where a minimum is wanted so that every word is a maximum over zero-initialised memory. A meeting is an
// arrival counter polled by one lane per workgroup with an atomic, BOUNDED by wall time: the grid is sized to be co-resident
// (at most one workgroup per CU), and should a workgroup nevertheless not arrive within kFusedPatienceTicks the call gives
// up (chunks = -1) and the host takes the launch-per-chunk path -- no wave can wait for ever.
// The state of the call after this one (the other parity block) is cleared here, by atomics as well.
constexpr int kFusedBlock = 512;
constexpr int kFusedMaxChunks = 64;
constexpr int kFusedMaxQueries = 8;
// A meeting takes ~5 us when every workgroup is resident (profiles/r03_twd_fused_latency.txt); the launch is only made when the occupancy
// query says the whole grid fits the device at once (fused_grid_fits), so a workgroup that has not arrived after 5 ms is one that another
// stream's kernels keep off the chip: the call then costs a fall-back to the launch-per-chunk form, not a quarter of a second.
constexpr unsigned long long kFusedPatienceTicks = 500000ull;       // wall_clock64() runs at 100 MHz: 5 ms
struct FusedSlot {
    unsigned long long vmin, best, cmin, cmax;    // ~orderable(min sum); ~((row << 32) | class); ~(ord(class) + 1); ord(class) + 1
};
struct FusedState {
    unsigned int ctr, done;
    unsigned long long pad;
    FusedSlot slot[kFusedMaxChunks];
    unsigned long long flag[256];                   // per workgroup: the number of the last meeting it has been released from (fused_meet)
};
__device__ __forceinline__ unsigned long long ord64(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double ord64_back(unsigned long long o) {
    const unsigned long long b = (o >> 63) ? (o & 0x7FFFFFFFFFFFFFFFull) : ~o;
    return __longlong_as_double((long long)b);
}
__device__ __forceinline__ unsigned long long atomic_max_read(unsigned long long* p, unsigned long long v) {
    return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// A post the other workgroups will read behind the next meeting: the RETURNING form -- once the value is back (every wave waits
// with vmcnt(0) in front of its workgroup's arrival) the maximum has been taken where every XCD's atomics are. The returned values are
// only parked in variables that are looked at behind the meeting: nothing waits for a single post.
__device__ __forceinline__ void post_max(unsigned long long* p, unsigned long long v, unsigned long long& ret) {
    ret = __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void keep_alive(unsigned long long a, unsigned long long b = 0, unsigned long long c = 0) { asm volatile("" ::"v"(a), "v"(b), "v"(c)); }
// A zero the compiler cannot see: an atomic maximum with a KNOWN zero is turned into an sc1 load, and what is wanted here is the
// value at the memory side.
__device__ __forceinline__ unsigned int opaque_zero() {
    unsigned int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return z;
}

constexpr int kMeetDirect = 16;
// One meeting of the `G` workgroups of a query. Thread 0 arrives (its posts have returned) with a returning add on `ctr`; the workgroup
// whose add completes `target` arrivals RELEASES the others by raising flag[0 .. G) to `gen` (one wave, 64 words per instruction);
// everybody else polls ITS OWN flag word -- with all of them polling the one counter the polls queue up behind each other at the memory
// side (196 workgroups: a meeting took 8-9 us against 4-5 us with 6). `after_arrival` runs in wave 0 between the arrival and the wait
// (requests that must not delay the arrival). The poll is bounded by wall time: false = the others did not come. Ends on a barrier.
template <typename F>
__device__ __forceinline__ bool fused_meet(unsigned int* ctr, unsigned long long* flags, int G, int b, unsigned int target, unsigned long long gen, bool wait,
                                           int* fail_s, F&& after_arrival) {
    if (threadIdx.x < 64) {
        const int lane = threadIdx.x;
        unsigned int seen = 0;
        if (lane == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            seen = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        }
        after_arrival();
        const bool direct = G <= kMeetDirect;                          // few workgroups: they poll the counter itself (one hop less)
        const int last = __shfl((int)(seen == target), 0, 64);
        if (last) {
            if (!direct)
                for (int i = lane; i < G; i += 64) (void)__hip_atomic_fetch_max(&flags[i], gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (lane == 0 && wait) {
            unsigned long long t0 = 0;
            for (unsigned int polls = 1;; ++polls) {
                if (direct ? __hip_atomic_fetch_max(ctr, opaque_zero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target
                           : __hip_atomic_fetch_max(&flags[b], (unsigned long long)opaque_zero(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gen) break;
                __builtin_amdgcn_s_sleep(1);
                if ((polls & 15u) == 0u) {                              // (the clock is a scalar memory read: not on every poll)
                    const unsigned long long now = wall_clock64();
                    if (!t0) t0 = now;
                    else if (now - t0 > kFusedPatienceTicks) { *fail_s = 1; break; }
                }
            }
        }
    }
    __syncthreads();
    return *fail_s == 0;
}

template <int METRIC, int T>
__global__ void __launch_bounds__(kFusedBlock) k_twd_prop_fused(const float4* __restrict__ gal4, int dp4, int n, int tiles,
                                                                 const int32_t* __restrict__ cls, const float* queries, int qstride,
                                                                 int reduced, int nchunks, double threshold /* 1/th */, FusedState* state,
                                                                 int parity, unsigned long long gen_base, int32_t* __restrict__ class_out,
                                                                 int32_t* __restrict__ unreliable_out, int32_t* __restrict__ chunks_out,
                                                                 int32_t* host_res, int host_stride, uint64_t* host_ticket, uint64_t ticket) {
    constexpr int kWaves = kFusedBlock / 64;
    __shared__ __attribute__((aligned(16))) float qs[kLastFeature];    // the query's 256 compared features: ONE read of (pinned host) memory per workgroup
    __shared__ unsigned long long red_key[kWaves];
    __shared__ unsigned int red_row[kWaves], red_cmin[kWaves], red_cmax[kWaves];
    __shared__ unsigned long long got[4];
    __shared__ int fail_s;
    const int q = blockIdx.y, b = blockIdx.x, G = gridDim.x, nq = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    FusedState* S = state + (size_t)q * 2 + parity;
    if (b == 0) {   // the next call's state (nothing of this launch touches it): all of it, whatever this call's queries and chunks
        constexpr int words = (int)(sizeof(FusedState) / 4);
        for (int qq = q; qq < kFusedMaxQueries; qq += nq) {
            unsigned int* other = (unsigned int*)(state + (size_t)qq * 2 + (parity ^ 1));
            for (int i = threadIdx.x; i < words; i += kFusedBlock) __hip_atomic_fetch_and(other + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    const float* qrow = queries + (size_t)q * qstride;
    if (threadIdx.x < kLastFeature) qs[threadIdx.x] = qrow[threadIdx.x];
    if (threadIdx.x == 0) fail_s = 0;
    const int W = G * kWaves, w = b * kWaves + wave;
    const int r4 = reduced >> 2;
    const float fcount = (float)reduced;                               // db_features.cpp:40
    double a[T];
    int cl[T];
    unsigned int alive = 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int t = w + i * W;
        const int64_t row = (int64_t)t * 64 + lane;
        a[i] = 0.0;                                                    // :210
        cl[i] = 0;
        if (t < tiles && row < n) { alive |= 1u << i; cl[i] = cls[row]; }   // :217
    }
    __syncthreads();
    // distances of chunk c for the rows still alive, added to their sums; the workgroup's smallest sum (and its first row) to
    // thread 0, which returns ~orderable (0 = no row below 100000) and the row
    // The first eight 16-byte pieces of the wave's first tile for the chunk AFTER the coming meeting are requested in front of that
    // meeting (whether the tile will still have a row alive is only known behind it): they arrive while the workgroups wait for each other.
    float4 pre[8];
    bool have_pre = false;
    auto prefetch = [&](int cn) {
        have_pre = cn < nchunks && w < tiles;
        if (have_pre) {
            const float4* tp = gal4 + ((size_t)w * dp4 + (size_t)cn * r4) * 64 + lane;
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (u < r4) pre[u] = tp[(size_t)u * 64];
        }
    };
    auto chunk = [&](int c, const float* qv, unsigned int& row_out) -> unsigned long long {
        double md = 100000.0;                                          // bestDist = 100000 per chunk (:225)
        int mrow = -1;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (!__builtin_amdgcn_ballot_w64((alive >> i) & 1u)) continue;      // wave-uniform: nothing of this tile is alive
            const int t = w + i * W;
            const float4* tp = gal4 + ((size_t)t * dp4 + (size_t)c * r4) * 64 + lane;
            float acc = 0.0f;
            for (int k0 = 0; k0 < r4; k0 += 8) {
                float4 g[8];
                if (i == 0 && k0 == 0 && have_pre) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = pre[u];
                } else {
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (k0 + u < r4) g[u] = tp[(size_t)(k0 + u) * 64];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (k0 + u < r4) {
                        const float4 l = *(const float4*)(qv + 4 * (k0 + u));
                        acc = fir::accum<METRIC>(acc, l.x, g[u].x);
                        acc = fir::accum<METRIC>(acc, l.y, g[u].y);
                        acc = fir::accum<METRIC>(acc, l.z, g[u].z);
                        acc = fir::accum<METRIC>(acc, l.w, g[u].w);
                    }
                }
            }
            if ((alive >> i) & 1u) {
                const double v = a[i] + (double)(acc / fcount);        // distances[j] += distance(...) (:243)
                a[i] = v;
                if (v < md) { md = v; mrow = t * 64 + lane; }         // rows ascend with i: strict '<' keeps the first (:248)
            }
        }
        unsigned long long key = mrow >= 0 ? ord64(md) : ~0ull;
        unsigned int row = (unsigned int)mrow;
        const unsigned long long wk = fir::wave_min_u64(key);
        row = key == wk ? row : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int o = __shfl_xor(row, off, 64);
            row = o < row ? o : row;
        }
        __syncthreads();
        if (lane == 0) { red_key[wave] = wk; red_row[wave] = row; }
        __syncthreads();
        unsigned long long bk = red_key[0];
        unsigned int br = red_row[0];
#pragma unroll
        for (int i = 1; i < kWaves; ++i)
            if (red_key[i] < bk || (red_key[i] == bk && red_row[i] < br)) { bk = red_key[i]; br = red_row[i]; }
        row_out = br;
        return br == 0xFFFFFFFFu ? 0ull : ~bk;
    };
    unsigned long long ret_a = 0, ret_b = 0;                            // (post_max)
    unsigned int my_row = 0xFFFFFFFFu;
    unsigned long long my_vmin = chunk(0, qs, my_row);
    if (threadIdx.x == 0 && my_vmin) post_max(&S->slot[0].vmin, my_vmin, ret_a);      // (returning: it has been performed when the wait below ends)
    int bestCls = -1, unreliable = 0, used = 0, failed = 0;
    bool has_best = false;
    for (int c = 0;; ++c) {
        // ---- meeting c: every workgroup of the query has posted chunk c's minimum and chunk c - 1's survivors ----
        // (the requests for chunk c + 1 go out in front of the wait; in thread 0's wave behind its arrival, which they must not delay;
        // after the last chunk only workgroup 0 has anything left to do)
        if (wave != 0) prefetch(c + 1);
        (void)fused_meet(&S->ctr, S->flag, G, b, (unsigned int)G * (unsigned int)(c + 1), gen_base + (unsigned long long)(c + 1), c < nchunks || b == 0, &fail_s,
                         [&]() { prefetch(c + 1); });
        keep_alive(ret_a, ret_b);
        if (c == nchunks && b != 0) return;
        if (fail_s) { failed = 1; break; }
        if (threadIdx.x < 4) {
            unsigned long long* p = threadIdx.x == 0 ? &S->slot[c < nchunks ? c : 0].vmin
                                  : threadIdx.x == 1 ? &S->slot[c > 0 ? c - 1 : 0].best
                                  : threadIdx.x == 2 ? &S->slot[c > 0 ? c - 1 : 0].cmin : &S->slot[c > 0 ? c - 1 : 0].cmax;
            got[threadIdx.x] = atomic_max_read(p, (unsigned long long)opaque_zero());
        }
        __syncthreads();
        const unsigned long long g_vmin = got[0], g_best = got[1], g_cmin = got[2], g_cmax = got[3];
        if (c > 0) {
            // chunk c - 1's bookkeeping (:248-281), its best row and its survivors now being complete
            ++used;
            if (g_best) {
                const unsigned long long kb = ~g_best;
                bestCls = (int)(unsigned int)kb;
                has_best = true;
            }
            const unsigned long long oc = (unsigned long long)((unsigned int)bestCls ^ 0x80000000u) + 1ull;
            const bool others = g_cmax != 0ull && !(g_cmax == oc && ~g_cmin == oc);      // a survivor of another class than the best row's
            if (!others) break;                                        // num_of_variants == 1 (:278)
            if (c == 1) unreliable = 1;                                // :280-281
            if (c == nchunks) break;
        }
        const bool has_c = g_vmin != 0ull;
        if (!has_best && !has_c) { ++used; break; }                    // bestInd == -1: the reference would index dbImages[-1]; answered -1, like the other forms
        const double best_d = has_c ? ord64_back(~g_vmin) : 100000.0;
        const double dist_threshold = best_d * threshold;              // :256
        // prune (:260-266); the class range of what survives; the first row at the minimum
        unsigned int cmin = 0xFFFFFFFFu, cmax = 0u;
        bool any = false;
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if ((alive >> i) & 1u) {
                if (a[i] > dist_threshold) alive &= ~(1u << i);
                else {
                    const unsigned int oc = (unsigned int)cl[i] ^ 0x80000000u;
                    cmin = oc < cmin ? oc : cmin;
                    cmax = oc > cmax ? oc : cmax;
                    any = true;
                }
            }
        }
        const bool wave_any = __builtin_amdgcn_ballot_w64(any) != 0;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int o1 = __shfl_xor(cmin, off, 64), o2 = __shfl_xor(cmax, off, 64);
            cmin = o1 < cmin ? o1 : cmin;
            cmax = o2 > cmax ? o2 : cmax;
        }
        if (lane == 0) { red_cmin[wave] = wave_any ? cmin : 0xFFFFFFFFu; red_cmax[wave] = cmax; red_row[wave] = wave_any ? 1u : 0u; }
        __syncthreads();
        if (threadIdx.x < 3) {
            unsigned int mn = 0xFFFFFFFFu, mx = 0u, anyw = 0u;
#pragma unroll
            for (int i = 0; i < kWaves; ++i) {
                mn = red_cmin[i] < mn ? red_cmin[i] : mn;
                mx = red_cmax[i] > mx ? red_cmax[i] : mx;
                anyw |= red_row[i];
            }
            if (threadIdx.x == 0) {
                if (has_c && my_vmin == g_vmin) post_max(&S->slot[c].best, ~(((unsigned long long)my_row << 32) | (unsigned int)cls[my_row]), ret_b);
            } else if (anyw) {
                if (threadIdx.x == 1) post_max(&S->slot[c].cmin, ~((unsigned long long)mn + 1ull), ret_b);
                else post_max(&S->slot[c].cmax, (unsigned long long)mx + 1ull, ret_b);
            }
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            my_vmin = chunk(c + 1, qs + (c + 1) * reduced, my_row);
            if (threadIdx.x == 0 && my_vmin) post_max(&S->slot[c + 1].vmin, my_vmin, ret_a);
        }
    }
    if (b != 0 || threadIdx.x != 0) return;
    const int cl_out = failed ? -1 : has_best ? bestCls : -1;
    const int used_out = failed ? -1 : used;
    class_out[q] = cl_out;
    unreliable_out[q] = unreliable;
    chunks_out[q] = used_out;
    if (host_res) {
        host_res[q] = cl_out;
        host_res[host_stride + q] = unreliable;
        host_res[2 * host_stride + q] = used_out;
        __threadfence_system();
        // the query whose workgroup 0 comes last publishes the ticket (every query's verdict is in host memory by then)
        FusedState* S0 = state + parity;
        if (__hip_atomic_fetch_add(&S0->done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned int)nq)
            __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ---- ConventionalTWDClassifier as ONE launch per call (few queries): k_twd_conv_fused ----
// Same idea as k_twd_prop_fused: both partial distances of a row ([0, reduced) and [reduced, 256)) stay in the registers of the lane
// that owns it; here workgroup b owns a CONTIGUOUS run of rows (tiles b * 8 T ...: wave w, slot i, lane = row order), because the
// reference's secondBestDist depends on the scan order (ImageTesting.cpp:123-125). Two meetings:
//   before the first   every workgroup posts: the packed (distance, row) key of its first minimum (one 64-bit atomic: the stage-1
//                      distances are floats), its own minimum into slot_m[b], the bit pattern of its smallest second-stage value,
//                      and (type 0) the posteriors max exp(-100 d) of the classes it saw (global 64-bit atomic max);
//   between them       everybody knows the best row r* and its class C*. secondBestDist is the distance of the LAST record (strict
//                      prefix minimum in row order) whose class is not C* -- every record behind it has class C*, so it is the value
//                      bestDist had at the last class change. A workgroup in front of r* posts the distance of its last LOCAL record of
//                      another class (prefix minimum over its own rows only) into slot_l[b]: if that one is not a global record,
//                      no earlier one of the workgroup is (records fall). Whoever holds the smallest second-stage value posts its row;
//   behind the second  workgroup 0 alone goes on: with P_b = min(100000, m_0 .. m_{b-1}) the last b whose slot_l[b] < P_b gives
//                      secondBestDist; (type 0) the five largest class posteriors; the reliability test; the second-stage row when it
//                      fails. It reads every word with an atomic exchange that leaves 0 behind: the state is clean for the next call.
constexpr int kConvMaxClasses = 7680;
struct ConvState {
    unsigned int ctr, pad;
    unsigned long long key1, v2min, v2row;           // ~key_pack(d1, row); ~ord64(second-stage value); ~row
    unsigned long long slot_m[256], slot_l[256];     // per workgroup: ~orderable(min d1); ~orderable(d1 of its last local record of another class)
    unsigned long long gprob[kConvMaxClasses];       // bit patterns of the class posteriors (non-negative doubles order like integers)
    unsigned long long flag[256];                    // fused_meet's release words: they only ever rise (the meeting numbers of a handle never repeat)
};
__device__ __forceinline__ unsigned long long xchg0(unsigned long long* p) {
    return __hip_atomic_exchange(p, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int METRIC, int T>
__global__ void __launch_bounds__(kFusedBlock) k_twd_conv_fused(const float4* __restrict__ gal4, int dp4, int n, int tiles, const int32_t* __restrict__ cls,
                                                                 const float* queries, int qstride, int reduced, int num_classes, int type,
                                                                 double threshold, ConvState* state, unsigned long long gen_base,
                                                                 int32_t* __restrict__ class_out, int32_t* __restrict__ unreliable_out, int32_t* host_res, int host_stride,
                                                                 uint64_t* host_ticket, uint64_t ticket, unsigned int* done_ctr) {
    constexpr int kWaves = kFusedBlock / 64;
    extern __shared__ __attribute__((aligned(16))) unsigned long long probabs[];   // type 0: num_classes bit patterns
    __shared__ __attribute__((aligned(16))) float qs[kLastFeature];
    __shared__ unsigned long long red_a[kWaves], red_b[kWaves];
    __shared__ unsigned int red_r[kWaves];
    __shared__ float red_f[kWaves], red_g[kWaves];
    __shared__ int red_i[kWaves];
    __shared__ unsigned long long got[4];
    __shared__ int fail_s;
    __shared__ DI red[kWaves];
    const int q = blockIdx.y, b = blockIdx.x, G = gridDim.x, nq = gridDim.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ConvState* S = state + q;
    const float* qrow = queries + (size_t)q * qstride;
    if (threadIdx.x < kLastFeature) qs[threadIdx.x] = qrow[threadIdx.x];
    if (threadIdx.x == 0) fail_s = 0;
    const bool with_prob = type == 0;
    if (with_prob)
        for (int c = threadIdx.x; c < num_classes; c += kFusedBlock) probabs[c] = 0ull;
    __syncthreads();
    // ---- both partial distances of this lane's rows (db_features.cpp:22-42 over [0, reduced) and [reduced, 256)) ----
    const int r4 = reduced >> 2, t4 = (kLastFeature - reduced) >> 2;
    const float f1 = (float)reduced, f2 = (float)(kLastFeature - reduced);
    float d1[T];
    double v2[T];
    int cl[T];
    unsigned int valid = 0;
    const int tile_base = (b * kWaves + wave) * T;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int t = tile_base + i;
        const int64_t row = (int64_t)t * 64 + lane;
        d1[i] = 0.0f; v2[i] = 0.0; cl[i] = 0;
        if (t >= tiles) continue;                                       // wave-uniform
        const float4* tp = gal4 + (size_t)t * dp4 * 64 + lane;
        float a1 = 0.0f, a2 = 0.0f;
        if (T == 1 && (r4 & 7) == 0) {
            // one tile per wave (with more, the tiles' request groups already overlap and the second register set only costs: 1M rows,
            // T = 8: 232 -> 365 us) and reduced_features_count a multiple of 32 (the reference's 32 / 64 / 128): the row's 64 pieces as eight request groups in
            // straight-line code, group g + 1 requested before group g is added up -- every wave of the chip asks for its 8 KiB at the same
            // moment, so without the overlap a group costs a memory latency PLUS its share of the transfer (3.5 us per 12.8 MB at
            // 100 000 rows). The first r4 / 8 groups belong to the first stage: `run` is handed to a1 at that border and starts again.
            const int nb1 = r4 >> 3;
            float run = 0.0f;
            float4 ga[8], gb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ga[u] = tp[(size_t)u * 64];
#pragma unroll
            for (int bi = 0; bi < 8; ++bi) {
                float4 (&cur)[8] = (bi & 1) ? gb : ga;
                float4 (&nxt)[8] = (bi & 1) ? ga : gb;
                if (bi + 1 < 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) nxt[u] = tp[(size_t)((bi + 1) * 8 + u) * 64];
                }
                a1 = bi == nb1 ? run : a1;
                run = bi == nb1 ? 0.0f : run;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 l = *(const float4*)(qs + 4 * (bi * 8 + u));
                    run = fir::accum<METRIC>(run, l.x, cur[u].x); run = fir::accum<METRIC>(run, l.y, cur[u].y);
                    run = fir::accum<METRIC>(run, l.z, cur[u].z); run = fir::accum<METRIC>(run, l.w, cur[u].w);
                }
            }
            a2 = run;
        } else {
        for (int k0 = 0; k0 < r4; k0 += 8) {
            float4 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + u < r4) g[u] = tp[(size_t)(k0 + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < r4) {
                    const float4 l = *(const float4*)(qs + 4 * (k0 + u));
                    a1 = fir::accum<METRIC>(a1, l.x, g[u].x); a1 = fir::accum<METRIC>(a1, l.y, g[u].y);
                    a1 = fir::accum<METRIC>(a1, l.z, g[u].z); a1 = fir::accum<METRIC>(a1, l.w, g[u].w);
                }
            }
        }
        for (int k0 = 0; k0 < t4; k0 += 8) {
            float4 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) if (k0 + u < t4) g[u] = tp[(size_t)(r4 + k0 + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if (k0 + u < t4) {
                    const float4 l = *(const float4*)(qs + 4 * (r4 + k0 + u));
                    a2 = fir::accum<METRIC>(a2, l.x, g[u].x); a2 = fir::accum<METRIC>(a2, l.y, g[u].y);
                    a2 = fir::accum<METRIC>(a2, l.z, g[u].z); a2 = fir::accum<METRIC>(a2, l.w, g[u].w);
                }
            }
        }
        }
        if (row < n) {
            valid |= 1u << i;
            d1[i] = a1 / f1;                                            // distances[j] (:117)
            const float tail = (a2 / f2) * (float)(kLastFeature - reduced);          // float * int -> float (:174)
            v2[i] = ((double)d1[i] * reduced + tail) / kLastFeature;    // :173-174
            cl[i] = cls[row];
        }
    }
    // ---- what the workgroup posts before the first meeting ----
    unsigned long long ret_a = 0, ret_b = 0, ret_c = 0, ret_d = 0;      // (post_max)
    unsigned long long k1 = ~0ull;                                      // first minimum below 100000 (:123: strict '<' from bestDist = 100000)
    double m2d = 100000.0;                                              // second stage: bestDist = 100000 (:168)
    unsigned int m2r = 0xFFFFFFFFu;
    float mloc = __builtin_huge_valf();
#pragma unroll
    for (int i = 0; i < T; ++i) {
        if (!((valid >> i) & 1u)) continue;
        const unsigned int row = (unsigned int)((tile_base + i) * 64 + lane);
        if ((double)d1[i] < 100000.0) { const unsigned long long k = fir::key_pack(d1[i], row); k1 = k < k1 ? k : k1; }
        mloc = d1[i] < mloc ? d1[i] : mloc;                            // (NaN never enters, as it never sets a record)
        if (v2[i] < m2d) { m2d = v2[i]; m2r = row; }
        if (with_prob) {
            const double probab = exp(-(double)d1[i] * 100);           // DIST_WEIGHT = 100 (:113,119)
            const int c = cl[i];
            if (c >= 0 && c < num_classes) atomicMax(&probabs[c], (unsigned long long)__double_as_longlong(probab));      // :120-121
        }
    }
    {
        const unsigned long long wk1 = fir::wave_min_u64(k1);
        unsigned long long k2 = m2r != 0xFFFFFFFFu ? ord64(m2d) : ~0ull;
        const unsigned long long wk2 = fir::wave_min_u64(k2);
        unsigned int r2 = k2 == wk2 ? m2r : 0xFFFFFFFFu;
        float wm = mloc;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int o = __shfl_xor(r2, off, 64);
            r2 = o < r2 ? o : r2;
            const float of = __shfl_xor(wm, off, 64);
            wm = of < wm ? of : wm;
        }
        if (lane == 0) { red_a[wave] = wk1; red_b[wave] = wk2; red_r[wave] = r2; red_f[wave] = wm; }
    }
    __syncthreads();
    unsigned long long my_k1 = red_a[0], my_k2 = red_b[0];
    unsigned int my_r2 = red_r[0];
    float my_m = red_f[0];
#pragma unroll
    for (int i = 1; i < kWaves; ++i) {
        my_k1 = red_a[i] < my_k1 ? red_a[i] : my_k1;
        if (red_b[i] < my_k2 || (red_b[i] == my_k2 && red_r[i] < my_r2)) { my_k2 = red_b[i]; my_r2 = red_r[i]; }
        my_m = red_f[i] < my_m ? red_f[i] : my_m;
    }
    if (threadIdx.x == 0) {
        if (my_k1 != ~0ull) post_max(&S->key1, ~my_k1, ret_a);
        if (my_r2 != 0xFFFFFFFFu) post_max(&S->v2min, ~my_k2, ret_b);
        if (my_m < __builtin_huge_valf()) post_max(&S->slot_m[b], ~(unsigned long long)fir::f32_orderable(my_m + 0.0f), ret_c);
    }
    if (with_prob)
        for (int c = threadIdx.x; c < num_classes; c += kFusedBlock)
            if (probabs[c]) post_max(&S->gprob[c], probabs[c], ret_d);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                    // every thread's posterior posts have returned ...
    __syncthreads();                                                   // ... before thread 0 arrives for the workgroup
    bool ok = fused_meet(&S->ctr, S->flag, G, b, (unsigned int)G, gen_base + 1ull, true, &fail_s, []() {});
    keep_alive(ret_a, ret_b, ret_c);
    keep_alive(ret_d);
    // ---- between the meetings: r*, C*; the last local record of another class; the second-stage row ----
    if (ok) {
        if (threadIdx.x < 2) got[threadIdx.x] = atomic_max_read(threadIdx.x == 0 ? &S->key1 : &S->v2min, (unsigned long long)opaque_zero());
        __syncthreads();
        const unsigned long long gk1 = got[0], gk2 = got[1];
        const bool has_best = gk1 != 0ull;
        const unsigned int rstar = has_best ? (unsigned int)(~gk1 & 0xFFFFFFFFull) : 0xFFFFFFFFu;
        if (has_best && type != 0) {
            const int cstar = cls[rstar];
            const unsigned int first_row = (unsigned int)(b * kWaves * T) * 64u;
            if (first_row < rstar) {                                    // (workgroups behind r* have no record in front of it)
                // wave-exclusive prefix of the waves' minima, then the scan in row order: tile by tile, lanes ascending
                float carry = __builtin_huge_valf();
                for (int w = 0; w < wave; ++w) carry = red_f[w] < carry ? red_f[w] : carry;
                float found_d = 0.0f;
                int found = 0;
#pragma unroll
                for (int i = 0; i < T; ++i) {
                    const unsigned int row = (unsigned int)((tile_base + i) * 64 + lane);
                    const bool live = ((valid >> i) & 1u) && d1[i] == d1[i];        // (a NaN distance never sets a record nor lowers bestDist)
                    const float dv = live ? d1[i] : __builtin_huge_valf();
                    float excl = __shfl_up(dv, 1, 64);                   // exclusive prefix minimum over the lanes in front
                    excl = lane == 0 ? __builtin_huge_valf() : excl;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const float o = __shfl_up(excl, off, 64);
                        if (lane >= off) excl = o < excl ? o : excl;
                    }
                    excl = carry < excl ? carry : excl;
                    const bool cand = live && dv < excl && cl[i] != cstar && row < rstar;
                    const unsigned long long bal = __builtin_amdgcn_ballot_w64(cand);
                    if (bal) {
                        const int last = 63 - __builtin_clzll(bal);
                        found_d = __shfl(dv, last, 64);
                        found = 1;
                    }
                    float tm = dv;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(tm, off, 64); tm = o < tm ? o : tm; }
                    carry = tm < carry ? tm : carry;
                }
                if (lane == 0) { red_g[wave] = found_d; red_i[wave] = found; }
            } else if (lane == 0) red_i[wave] = 0;
            __syncthreads();
            if (threadIdx.x == 0) {
                int have = 0;
                float dl = 0.0f;
                for (int w = 0; w < kWaves; ++w) if (red_i[w]) { have = 1; dl = red_g[w]; }      // the last wave that found one
                if (have) post_max(&S->slot_l[b], ~(unsigned long long)fir::f32_orderable(dl + 0.0f), ret_a);
            }
        }
        if (threadIdx.x == 0 && gk2 != 0ull && my_r2 != 0xFFFFFFFFu && ~my_k2 == gk2) post_max(&S->v2row, ~(unsigned long long)my_r2, ret_b);
        ok = fused_meet(&S->ctr, S->flag, G, b, 2u * (unsigned int)G, gen_base + 2ull, b == 0, &fail_s, []() {});
        keep_alive(ret_a, ret_b);
    }
    if (b != 0) return;
    // ---- behind the second meeting: workgroup 0 decides ----
    int cls_final = -1, reliable = 0, failed = ok ? 0 : 1;
    if (ok) {
        __shared__ unsigned long long sm_s[256], sl_s[256];
        __shared__ double second_s;
        if (threadIdx.x < 3) got[threadIdx.x] = xchg0(threadIdx.x == 0 ? &S->key1 : threadIdx.x == 1 ? &S->v2min : &S->v2row);
        for (int i = threadIdx.x; i < G; i += kFusedBlock) { sm_s[i] = xchg0(&S->slot_m[i]); sl_s[i] = xchg0(&S->slot_l[i]); }
        if (with_prob)
            for (int c = threadIdx.x; c < num_classes; c += kFusedBlock) probabs[c] = xchg0(&S->gprob[c]);
        __syncthreads();
        const unsigned long long gk1 = got[0];
        const bool has_best = gk1 != 0ull;
        const unsigned int rstar = has_best ? (unsigned int)(~gk1 & 0xFFFFFFFFull) : 0xFFFFFFFFu;
        const double best_d = has_best ? (double)fir::f32_from_orderable((uint32_t)(~gk1 >> 32)) : 100000.0;
        if (threadIdx.x == 0) {
            double second = 100000.0;                                   // secondBestDist = 100000 (:111)
            float P = 100000.0f;
            for (int i = 0; i < G; ++i) {
                if (sl_s[i]) {
                    const float dl = fir::f32_from_orderable((uint32_t)~sl_s[i]);
                    if (dl < P) second = (double)dl;
                }
                if (sm_s[i]) { const float m = fir::f32_from_orderable((uint32_t)~sm_s[i]); P = m < P ? m : P; }
            }
            second_s = second;
        }
        __syncthreads();
        if (has_best) {
            if (type == 0) {
                // sum of the 5 largest class posteriors (:141-146), taken in descending order
                double sum = 0.0;
                for (int r = 0; r < 5; ++r) {
                    DI m;
                    m.d = __builtin_huge_val();
                    m.i = -1;
                    for (int c = threadIdx.x; c < num_classes; c += kFusedBlock) {
                        const double v = __longlong_as_double((long long)probabs[c]);
                        if (probabs[c] != ~0ull && (-v < m.d)) { m.d = -v; m.i = c; }
                    }
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) {
                        DI o;
                        o.d = __shfl_xor(m.d, off, 64);
                        o.i = __shfl_xor(m.i, off, 64);
                        if (o.d < m.d || (o.d == m.d && (unsigned)o.i < (unsigned)m.i)) m = o;
                    }
                    __syncthreads();
                    if (lane == 0) red[wave] = m;
                    __syncthreads();
                    DI w = red[0];
#pragma unroll
                    for (int i = 1; i < kWaves; ++i)
                        if (red[i].d < w.d || (red[i].d == w.d && (unsigned)red[i].i < (unsigned)w.i)) w = red[i];
                    sum += -w.d;
                    __syncthreads();
                    if (threadIdx.x == 0 && w.i >= 0) probabs[w.i] = ~0ull;             // taken
                    __syncthreads();
                }
                const double max_probab = exp(-best_d * 100) / sum;                     // :130,147
                reliable = max_probab > threshold;                                      // :148
            } else if (type == 1) {
                reliable = (second_s - best_d) > threshold;                             // :158
            } else {
                reliable = (best_d / second_s) < threshold;                             // :161
            }
        }
        cls_final = has_best ? cls[rstar] : -1;
        if (!reliable) {                                                                // second stage (:165-180)
            const unsigned long long gr = got[2];
            cls_final = (got[1] != 0ull && gr != 0ull) ? cls[(unsigned int)~gr] : -1;
        }
    }
    if (threadIdx.x != 0) return;
    (void)__hip_atomic_exchange(&S->ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);        // everybody has arrived twice: nobody reads it any more
    class_out[q] = cls_final;
    unreliable_out[q] = failed ? 2 : reliable ? 0 : 1;                  // 2: the workgroups did not meet in time -- the host takes the other form
    if (host_res) {
        host_res[q] = cls_final;
        host_res[host_stride + q] = failed ? 2 : reliable ? 0 : 1;
        __threadfence_system();
        if (__hip_atomic_fetch_add(done_ctr, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) + 1u == (unsigned int)nq) {
            (void)__hip_atomic_exchange(done_ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
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

// The verdicts of a batch go to pinned host memory, then the ticket the host is spinning on (fir_gallery_wait_ticket_).
__global__ void __launch_bounds__(256) k_twd_publish(const int32_t* __restrict__ res, int count, int32_t* __restrict__ host_res,
                                                     uint64_t* __restrict__ host_ticket, uint64_t ticket) {
    for (int i = threadIdx.x; i < count; i += 256) host_res[i] = res[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(host_ticket, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" {

// The hand-rolled grid meetings of the fused kernels need every workgroup of the launch resident at the same time: checked against the
// occupancy query before the launch (MI355X_MICROARCH.md: the API can read one block per CU high near register-file edges, so one block of
// margin is kept whenever more than one per CU is counted on), and bounded by kFusedPatienceTicks inside the kernel whatever else runs.
static bool fused_grid_fits(const void* fn, int block, size_t dyn_lds, int workgroups, int cus) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, block, dyn_lds) != hipSuccess) { (void)hipGetLastError(); return false; }
    if (per_cu < 1) return false;
    const int safe = per_cu > 1 ? per_cu - 1 : 1;
    return workgroups <= safe * std::max(cus, 1);
}

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
    TWD_SLOT(dres, 3, (size_t)2 * kBatch * 4);                 // class[kBatch], unreliable[kBatch]
    int32_t* dcls = dres.as<int32_t>();
    int32_t* dunrel = dcls + kBatch;
    // galleries beyond a few spans: the rows of every query are split over `nseg` workgroups (see k_twd_conv_stage1)
    const int nseg = n > 4 * kSpan ? std::min(256, (n + 2 * kSpan - 1) / (2 * kSpan)) : 1;
    const int seg_rows = nseg > 1 ? ((n + nseg - 1) / nseg + kSpan - 1) / kSpan * kSpan : n;
    const int nseg_eff = nseg > 1 ? (n + seg_rows - 1) / seg_rows : 1;
    // (the distance tables and segment records of the launch-per-stage form are taken from the handle's scratch when that form runs)
    Slot d1, dpart;
    DI *part1 = nullptr, *part2 = nullptr;
    S1Chg* chg = nullptr;
    unsigned long long* gprob = nullptr;
    auto stage_scratch = [&]() -> int {
        int rc2;
        if ((rc2 = fir_gallery_scratch_(g, 1, std::max<size_t>((size_t)2 * batch * std::max(n, 1) * 4, 16), &d1.p))) return rc2;    // [0, reduced) distances of the batch, then [reduced, 256)
        if ((rc2 = fir_gallery_scratch_(g, 7, std::max<size_t>((size_t)batch * nseg_eff * (sizeof(DI) * 2 + sizeof(S1Chg)) + (size_t)batch * num_classes * 8, 16), &dpart.p)))
            return rc2;
        part1 = dpart.as<DI>();
        part2 = part1 + (size_t)batch * nseg_eff;
        chg = (S1Chg*)(part2 + (size_t)batch * nseg_eff);
        gprob = (unsigned long long*)(chg + (size_t)batch * nseg_eff);
        return FIR_OK;
    };
    // Few queries: ONE launch per internal batch (k_twd_conv_fused), as for the proposed classifier (FIR_TWD_FUSED).
    const char* fenv = fir_knob_("FIR_TWD_FUSED");
    const int fmode = fenv ? std::atoi(fenv) : 1;
    const int64_t tiles64 = ((int64_t)n + 63) / 64;
    bool fused = fmode != 0 && n > 0 && (fmode == 2 || qb <= kFusedMaxQueries) && (v.metric == 0 || v.metric == 1) && reduced_features_count % 4 == 0 &&
                 num_classes <= 4096;                                   // (the posteriors of a workgroup sit in LDS next to 6 KiB of static tables)
    const int fq = std::min(qb, kFusedMaxQueries);
    int fG = 0, fT = 0;
    if (fused) {
        for (int t : {1, 2, 4, 8, 16}) {
            const int64_t g_need = (tiles64 + 8 * t - 1) / (8 * t);
            if (g_need <= std::min(256, std::max(1, v.cus / fq))) { fT = t; fG = (int)g_need; break; }
        }
        // every query of a fused launch reads the rows for itself: beyond one tile per wave that costs more than the launches it saves
        // (100 000 x 512, 8 queries: 146 against 96 us), so several queries go this way only while every wave has a single tile
        if (!fT || (fq > 1 && fT > 1)) fused = false;
        if (fused) {      // every workgroup of the launch must be resident at once: the occupancy query decides, not an assumption
            const void* f16 = v.metric == 0 ? (fT == 1 ? (const void*)k_twd_conv_fused<fir::kL2, 1> : fT == 2 ? (const void*)k_twd_conv_fused<fir::kL2, 2>
                                                     : fT == 4 ? (const void*)k_twd_conv_fused<fir::kL2, 4> : fT == 8 ? (const void*)k_twd_conv_fused<fir::kL2, 8>
                                                                                                                      : (const void*)k_twd_conv_fused<fir::kL2, 16>)
                                            : (fT == 1 ? (const void*)k_twd_conv_fused<fir::kChi2, 1> : fT == 2 ? (const void*)k_twd_conv_fused<fir::kChi2, 2>
                                                     : fT == 4 ? (const void*)k_twd_conv_fused<fir::kChi2, 4> : fT == 8 ? (const void*)k_twd_conv_fused<fir::kChi2, 8>
                                                                                                                        : (const void*)k_twd_conv_fused<fir::kChi2, 16>);
            if (!fused_grid_fits(f16, kFusedBlock, type == 0 ? (size_t)num_classes * 8 : 8, fG * fq, v.cus)) fused = false;
        }
    }
    const int fbatch = fused ? kFusedMaxQueries : batch;
    for (int q0 = 0; q0 < qb; q0 += fbatch) {
        const int nq = std::min(fbatch, qb - q0);
        int32_t h_res[2 * kBatch];
        bool answered = false;
        if (fused) {
            void* pin_base = nullptr; size_t pin_cap = 0; uint64_t* pin_res = nullptr;
            const bool pinned = fir_gallery_pin_(g, &pin_base, &pin_cap, &pin_res) == FIR_OK && (size_t)nq * v.d * 4 <= pin_cap;
            const float* qsrc = dq.as<float>();
            if (pinned) { std::memcpy(pin_base, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4); qsrc = (const float*)pin_base; }
            else TWD_HIP(hipMemcpyAsync(dq.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
            TWD_SLOT(cst, 16, (size_t)kFusedMaxQueries * sizeof(ConvState) + 64);        // (a fresh slot reads as zeros; the kernel leaves it so)
            ConvState* cstate = cst.as<ConvState>();
            unsigned int* done_ctr = (unsigned int*)(cstate + kFusedMaxQueries);
            const void* gal4 = nullptr;
            int dp4 = 0;
            if ((rc = fir_gallery_tiled_(g, &gal4, &dp4))) return twd_fail(rc, "no tiled gallery");
            const uint64_t ticket = pinned ? fir_gallery_next_ticket_(g) : 0;
            typedef void (*conv_fn)(const float4*, int, int, int, const int32_t*, const float*, int, int, int, int, double, ConvState*, unsigned long long, int32_t*,
                                    int32_t*, int32_t*, int, uint64_t*, uint64_t, unsigned int*);
            const unsigned long long gen_base = (unsigned long long)(fir_gallery_next_counter_(g, 1) + 1) << 7;      // meeting numbers that never repeat
            conv_fn fn = nullptr;
#define FIR_CONV_PICK(M)                                                                                               \
    fn = fT == 1 ? k_twd_conv_fused<M, 1> : fT == 2 ? k_twd_conv_fused<M, 2> : fT == 4 ? k_twd_conv_fused<M, 4>        \
       : fT == 8 ? k_twd_conv_fused<M, 8> : k_twd_conv_fused<M, 16>
            if (v.metric == 0) { FIR_CONV_PICK(fir::kL2); } else { FIR_CONV_PICK(fir::kChi2); }
#undef FIR_CONV_PICK
            const size_t flds = type == 0 ? (size_t)num_classes * 8 : 8;      // <= 32 KiB
            hipLaunchKernelGGL(fn, dim3(fG, nq), dim3(kFusedBlock), flds, v.stream, (const float4*)gal4, dp4, n, (int)tiles64, v.cls, qsrc, v.d,
                               reduced_features_count, num_classes, type, threshold, cstate, gen_base, dcls, dunrel, pinned ? (int32_t*)pin_res : (int32_t*)nullptr, kBatch,
                               pinned ? pin_res + kBatch : (uint64_t*)nullptr, ticket, done_ctr);
            TWD_HIP(hipGetLastError());
            if (pinned) {
                if ((rc = fir_gallery_wait_ticket_(g, pin_res + kBatch, ticket))) return rc;
                std::memcpy(h_res, pin_res, sizeof(h_res));
            } else {
                TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
                TWD_HIP(hipStreamSynchronize(v.stream));
            }
            answered = true;
            for (int i = 0; i < nq; ++i) answered = answered && h_res[kBatch + i] != 2;        // 2: the workgroups did not meet in time
            if (!answered) {
                TWD_HIP(hipStreamSynchronize(v.stream));
                TWD_HIP(hipMemsetAsync(cst.p, 0, (size_t)kFusedMaxQueries * sizeof(ConvState) + 64, v.stream));      // whatever the launch left behind
            }
        }
        if (answered) {
        } else if (n == 0) {
            for (int i = 0; i < nq; ++i) { h_res[i] = -1; h_res[kBatch + i] = 1; }
        } else {
            if ((rc = stage_scratch())) return rc;
            // both stages are queued back to back -- the second one decides on the device which queries it concerns -- and
            // the verdicts come back with ONE copy and ONE synchronisation per batch
            // small batches: the kernels read the queries from pinned host memory and the verdicts come back through it, with
            // a ticket instead of a stream synchronisation (no copy engine on either side)
            void* pin_base = nullptr; size_t pin_cap = 0; uint64_t* pin_res = nullptr;
            const bool pinned = fir_gallery_pin_(g, &pin_base, &pin_cap, &pin_res) == FIR_OK && (size_t)nq * v.d * 4 <= pin_cap;
            const float* qsrc = dq.as<float>();
            if (pinned) { std::memcpy(pin_base, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4); qsrc = (const float*)pin_base; }
            else TWD_HIP(hipMemcpyAsync(dq.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
            // both partial distances from one pass over features [0, 256); every stage is queued back to back
            if ((rc = fir_split_distances_dev_(g, qsrc, nq, reduced_features_count, kLastFeature, d1.as<float>(), v.stream))) return rc;
            const float* d2 = d1.as<float>() + (size_t)nq * n;
            const size_t plds = (size_t)num_classes * 8;
            // a one-query call: the deciding workgroup writes the verdict to pinned host memory and the ticket itself
            const bool self_publish = pinned && nq == 1;
            const uint64_t ticket = pinned ? fir_gallery_next_ticket_(g) : 0;
            int32_t* hres = self_publish ? (int32_t*)pin_res : (int32_t*)nullptr;
            uint64_t* hticket = self_publish ? pin_res + kBatch : (uint64_t*)nullptr;
            if (nseg_eff == 1) {
                hipLaunchKernelGGL(k_twd_conv_stage1<kS1Single>, dim3(nq), dim3(kBlock), plds, v.stream, d1.as<float>(), v.cls, n, num_classes, type,
                                   threshold, dcls, dunrel, (DI*)nullptr, (S1Chg*)nullptr, (unsigned long long*)nullptr, n, d2,
                                   reduced_features_count, (DI*)nullptr, hres, kBatch, hticket, ticket);
            } else {
                if (type == 0) TWD_HIP(hipMemsetAsync(gprob, 0, (size_t)nq * num_classes * 8, v.stream));
                hipLaunchKernelGGL(k_twd_conv_stage1<kS1Part>, dim3(nseg_eff, nq), dim3(kBlock), plds, v.stream, d1.as<float>(), v.cls, n, num_classes,
                                   type, threshold, dcls, dunrel, part1, chg, gprob, seg_rows, d2, reduced_features_count, part2, (int32_t*)nullptr, 0,
                                   (uint64_t*)nullptr, (uint64_t)0);
                hipLaunchKernelGGL(k_twd_conv_stage1<kS1Records>, dim3(nseg_eff, nq), dim3(kBlock), plds, v.stream, d1.as<float>(), v.cls, n,
                                   num_classes, type, threshold, dcls, dunrel, part1, chg, gprob, seg_rows, d2, reduced_features_count, part2,
                                   (int32_t*)nullptr, 0, (uint64_t*)nullptr, (uint64_t)0);
                hipLaunchKernelGGL(k_twd_conv_stage1<kS1Final>, dim3(nq), dim3(kBlock), plds, v.stream, d1.as<float>(), v.cls, n, num_classes, type,
                                   threshold, dcls, dunrel, part1, chg, gprob, seg_rows, d2, reduced_features_count, part2, hres, kBatch, hticket,
                                   ticket);
            }
            TWD_HIP(hipGetLastError());
            if (pinned) {
                if (!self_publish) {
                    hipLaunchKernelGGL(k_twd_publish, dim3(1), dim3(256), 0, v.stream, dcls, 2 * kBatch, (int32_t*)pin_res, pin_res + kBatch, ticket);
                    TWD_HIP(hipGetLastError());
                }
                if ((rc = fir_gallery_wait_ticket_(g, pin_res + kBatch, ticket))) return rc;
                std::memcpy(h_res, pin_res, sizeof(h_res));
            } else {
                TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
                TWD_HIP(hipStreamSynchronize(v.stream));
            }
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
    // Few queries: ONE launch per internal batch (k_twd_prop_fused). FIR_TWD_FUSED=0 never, 2 = whatever the batch (tests).
    const char* fenv = fir_knob_("FIR_TWD_FUSED");
    const int fmode = fenv ? std::atoi(fenv) : 1;
    const int64_t tiles = ((int64_t)n + 63) / 64;
    bool fused = fmode != 0 && n > 0 && (fmode == 2 || qb <= kFusedMaxQueries) && (v.metric == 0 || v.metric == 1) &&
                 reduced_features_count % 4 == 0 && reduced_features_count <= 128 && nchunks <= kFusedMaxChunks;
    const int fq = std::min(qb, kFusedMaxQueries);                       // queries per fused launch
    int fG = 0, fT = 0;
    if (fused) {
        // at most one workgroup per CU over all the queries of a launch: the workgroups of a query wait for each other
        fG = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1, v.cus / fq), (tiles + 7) / 8));
        const int64_t need = (tiles + (int64_t)fG * 8 - 1) / ((int64_t)fG * 8);
        fT = need <= 1 ? 1 : need <= 2 ? 2 : need <= 4 ? 4 : need <= 8 ? 8 : need <= 16 ? 16 : 0;
        if (!fT) fused = false;
        if (fused) {      // every workgroup of the launch must be resident at once: the occupancy query decides, not an assumption
            const void* f16 = v.metric == 0 ? (fT == 1 ? (const void*)k_twd_prop_fused<fir::kL2, 1> : fT == 2 ? (const void*)k_twd_prop_fused<fir::kL2, 2>
                                                     : fT == 4 ? (const void*)k_twd_prop_fused<fir::kL2, 4> : fT == 8 ? (const void*)k_twd_prop_fused<fir::kL2, 8>
                                                                                                                      : (const void*)k_twd_prop_fused<fir::kL2, 16>)
                                            : (fT == 1 ? (const void*)k_twd_prop_fused<fir::kChi2, 1> : fT == 2 ? (const void*)k_twd_prop_fused<fir::kChi2, 2>
                                                     : fT == 4 ? (const void*)k_twd_prop_fused<fir::kChi2, 4> : fT == 8 ? (const void*)k_twd_prop_fused<fir::kChi2, 8>
                                                                                                                        : (const void*)k_twd_prop_fused<fir::kChi2, 16>);
            if (!fused_grid_fits(f16, kFusedBlock, 0, fG * fq, v.cus)) fused = false;
        }
    }
    const int batch = fused ? 8 : std::min(batch_for(n, (size_t)nchunks * 4 + 9, (size_t)1 << 30), std::max(8, (qb + 7) / 8 * 8));
    TWD_SLOT(dq, 0, (size_t)batch * v.d * 4);
    TWD_SLOT(dres, 3, (size_t)3 * kBatch * 4);                 // class, unreliable, chunks
    int32_t* dcls = dres.as<int32_t>();
    int32_t* dunrel = dcls + kBatch;
    int32_t* dchunks = dcls + 2 * kBatch;
    // large galleries: the rows of every query are split over `nseg` workgroups, one launch per chunk
    const int nseg_want = n > 16384 ? std::min(256, (n + 8191) / 8192) : 1;
    const int seg_rows = nseg_want > 1 ? ((n + nseg_want - 1) / nseg_want + 255) / 256 * 256 : n;
    const int nseg = nseg_want > 1 ? (n + seg_rows - 1) / seg_rows : 1;
    for (int q0 = 0; q0 < qb; q0 += batch) {
        const int nq = std::min(batch, qb - q0);
        int32_t h_res[3 * kBatch];
        int32_t* h_cls = h_res;
        int32_t* h_unrel = h_res + kBatch;
        int32_t* h_chunks = h_res + 2 * kBatch;
        if (n == 0) {
            for (int i = 0; i < nq; ++i) { h_cls[i] = -1; h_unrel[i] = 0; h_chunks[i] = 0; }
        } else {
            void* pin_base = nullptr; size_t pin_cap = 0; uint64_t* pin_res = nullptr;      // as in fir_twd_conventional
            const bool pinned = fir_gallery_pin_(g, &pin_base, &pin_cap, &pin_res) == FIR_OK && (size_t)nq * v.d * 4 <= pin_cap;
            const float* qsrc = dq.as<float>();
            if (pinned) { std::memcpy(pin_base, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4); qsrc = (const float*)pin_base; }
            else TWD_HIP(hipMemcpyAsync(dq.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
            bool answered = false;
            if (fused) {
                TWD_SLOT(fst, 2, (size_t)2 * kFusedMaxQueries * sizeof(FusedState));
                const void* gal4 = nullptr;
                int dp4 = 0;
                if ((rc = fir_gallery_tiled_(g, &gal4, &dp4))) return twd_fail(rc, "no tiled gallery");
                const uint64_t call_no = fir_gallery_next_counter_(g, 0);
                const int parity = (int)(call_no & 1);
                const unsigned long long gen_base = (unsigned long long)(call_no + 1) << 7;                        // meeting numbers that never repeat
                const uint64_t ticket = pinned ? fir_gallery_next_ticket_(g) : 0;
                typedef void (*fused_fn)(const float4*, int, int, int, const int32_t*, const float*, int, int, int, double, FusedState*, int, unsigned long long,
                                         int32_t*, int32_t*, int32_t*, int32_t*, int, uint64_t*, uint64_t);
                fused_fn fn = nullptr;
#define FIR_FUSED_PICK(M)                                                                                              \
    fn = fT == 1 ? k_twd_prop_fused<M, 1> : fT == 2 ? k_twd_prop_fused<M, 2> : fT == 4 ? k_twd_prop_fused<M, 4>        \
       : fT == 8 ? k_twd_prop_fused<M, 8> : k_twd_prop_fused<M, 16>
                if (v.metric == 0) { FIR_FUSED_PICK(fir::kL2); } else { FIR_FUSED_PICK(fir::kChi2); }
#undef FIR_FUSED_PICK
                hipLaunchKernelGGL(fn, dim3(fG, nq), dim3(kFusedBlock), 0, v.stream, (const float4*)gal4, dp4, n, (int)tiles, v.cls, qsrc, v.d,
                                   reduced_features_count, nchunks, 1.0 / threshold, fst.as<FusedState>(), parity, gen_base, dcls, dunrel, dchunks,
                                   pinned ? (int32_t*)pin_res : (int32_t*)nullptr, kBatch, pinned ? pin_res + 2 * kBatch : (uint64_t*)nullptr, ticket);
                TWD_HIP(hipGetLastError());
                if (pinned) {
                    if ((rc = fir_gallery_wait_ticket_(g, pin_res + 2 * kBatch, ticket))) return rc;
                    std::memcpy(h_res, pin_res, sizeof(h_res));
                } else {
                    TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
                    TWD_HIP(hipStreamSynchronize(v.stream));
                }
                answered = true;
                for (int i = 0; i < nq; ++i) answered = answered && h_chunks[i] >= 0;      // -1: the workgroups did not meet in time
                if (!answered) TWD_HIP(hipStreamSynchronize(v.stream));
            }
            if (!answered) {
                const int ob = std::max(8, (nq + 7) / 8 * 8);
                TWD_SLOT(cd, 4, (size_t)nchunks * ob * std::max(n, 1) * 4);
                TWD_SLOT(acc, 5, (size_t)ob * std::max(n, 1) * 8);
                TWD_SLOT(alive, 6, (size_t)ob * std::max(n, 1));
                TWD_SLOT(pws, 7, (size_t)2 * ob * sizeof(PropState) + (size_t)nchunks * ob * sizeof(int) + (size_t)nchunks * ob * nseg * sizeof(DI) + 64);
                DI* ppart = pws.as<DI>();
                PropState* pstate = (PropState*)(ppart + (size_t)nchunks * ob * nseg);
                int* pcnt = (int*)(pstate + 2 * ob);
                // all chunk distances cd[c][slot][n] from ONE pass over features [0, 256)
                if ((rc = fir_subrange_distances_dev_(g, qsrc, nq, 0, kLastFeature, reduced_features_count, cd.as<float>(), v.stream))) return rc;
                const bool self_publish = pinned && nq == 1 && nseg == 1;    // the deciding workgroup writes verdict and ticket itself
                const uint64_t ticket = pinned ? fir_gallery_next_ticket_(g) : 0;
                if (nseg == 1) {
                    hipLaunchKernelGGL(k_twd_proposed, dim3(nq), dim3(kBlock), 0, v.stream, cd.as<float>(), nq, nchunks, acc.as<double>(),
                                       alive.as<uint8_t>(), v.cls, n, 1.0 / threshold, dcls, dunrel, dchunks,
                                       self_publish ? (int32_t*)pin_res : (int32_t*)nullptr, kBatch,
                                       self_publish ? pin_res + 2 * kBatch : (uint64_t*)nullptr, ticket);
                } else {
                    // nchunks + 1 launches + the finish (three launches per chunk before); the two state buffers swap roles every launch
                    hipLaunchKernelGGL(k_twd_prop_init, dim3((2 * ob + 63) / 64), dim3(64), 0, v.stream, pstate, 2 * ob);
                    TWD_HIP(hipMemsetAsync(pcnt, 0, (size_t)nchunks * nq * sizeof(int), v.stream));
                    for (int c = 0; c <= nchunks; ++c)
                        hipLaunchKernelGGL(k_twd_prop_chunk, dim3(nseg, nq), dim3(kBlock), 0, v.stream, cd.as<float>(), nq, c, nchunks, acc.as<double>(),
                                           alive.as<uint8_t>(), v.cls, n, seg_rows, 1.0 / threshold, pstate + (size_t)((c + 1) & 1) * ob,
                                           pstate + (size_t)(c & 1) * ob, ppart, pcnt);
                    hipLaunchKernelGGL(k_twd_prop_finish, dim3((nq + 63) / 64), dim3(64), 0, v.stream, nq, nchunks, nseg, ppart, pcnt, v.cls,
                                       pstate + (size_t)(nchunks & 1) * ob, dcls, dunrel, dchunks);
                }
                TWD_HIP(hipGetLastError());
                if (pinned) {
                    if (!self_publish) {
                        hipLaunchKernelGGL(k_twd_publish, dim3(1), dim3(256), 0, v.stream, dcls, 3 * kBatch, (int32_t*)pin_res, pin_res + 2 * kBatch, ticket);
                        TWD_HIP(hipGetLastError());
                    }
                    if ((rc = fir_gallery_wait_ticket_(g, pin_res + 2 * kBatch, ticket))) return rc;
                    std::memcpy(h_res, pin_res, sizeof(h_res));
                } else {
                    TWD_HIP(hipMemcpyAsync(h_res, dres.p, sizeof(h_res), hipMemcpyDeviceToHost, v.stream));
                    TWD_HIP(hipStreamSynchronize(v.stream));
                }
            }
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
