// fir_dem.hip -- the pivot table of DirectedEnumeration's constructor (qt_cpp/ann.cpp:302-331, PIVOT build) on gfx950.
//
// The reference walks its pivots one after another: a full gallery scan against pivot ii, then a choice of pivot ii+1
// as the row farthest (in summed distance) from all pivots so far. The choice depends on the scan before it, so the
// work is n_pivots dependent gallery passes of ONE query each -- HBM-bound (n*d*4 bytes per pivot), served by the
// library's range-distance kernel in the reference's arithmetic order. Everything stays on the device between
// pivots (no host round trip per pivot):
//   k_dem_gather   pivot row (tiled gallery, fir_kernels.h layout) -> dense query vector
//   range scan     table[ii][j] = distance(row j, pivot)                         (fir_range_distances_dev)
//   k_dem_step     far[j] (double, running; -1000000 restart at a pivot, :313-318), per-block partials of
//                  "first row with the largest far > 0" (:319-322) and of the minimum distance to another class (:309-311)
//   k_dem_pick     one workgroup folds the partials, writes min_other[ii] and pivots[ii+1]
// The running far[j] is the reference's inner `ind` loop evaluated incrementally: the same additions in the same order.
//
// Query time (DirectedEnumeration::recognize, ann.cpp:411-507) is a sequential, early-exit walk that stays on the
// host (host/fir_classifiers.cpp); its two data-parallel pieces are here:
//   k_dem_lik      likelihoods[nu] += (dist(query, pivot i) - table[i][nu])^2 for the <= 32 pivots kept (:437-446):
//                  one lane per gallery row, the pivots in order, float adds in the reference's order; n*P*4 bytes of
//                  table per batch of 8 queries instead of n*d*4 bytes of gallery
//   k_rows_dist    distance(query, row) for a per-query list of candidate rows (CHECK_FOR_BEST_DIST, :389-399)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <unordered_map>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_common.h"
#include "fir_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kMaxBlocks = 1024;

thread_local char g_dem_err[512];
int dem_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_dem_err, sizeof(g_dem_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_dem_err);
    return code;
}
#define DEM_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return dem_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,        \
                                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct Part {       // one block's partial result
    double far;     // largest far sum > 0 seen (0 = none)
    int32_t row;    // first row that reached it, -1 = none
    float min_other;
};

// pivots[ii] -> q[0..d). A pivot of -1 (the reference's "no row has a positive far sum", which it would crash on)
// yields a zero vector; the host reports the truncation.
__global__ void __launch_bounds__(kBlock) k_dem_gather(const float4* __restrict__ gal4, int dp4, int d, const int32_t* __restrict__ pivots,
                                                       int ii, float* __restrict__ q) {
    const int piv = pivots[ii];
    for (int c = blockIdx.x * kBlock + threadIdx.x; c < dp4; c += gridDim.x * kBlock) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (piv >= 0) v = gal4[((size_t)(piv >> 6) * dp4 + c) * 64 + (piv & 63)];
        const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c * 4 + k < d) q[c * 4 + k] = e[k];
    }
}

// (far desc, row asc) -- row -1 compares as the largest unsigned, so "none" loses every tie
__device__ __forceinline__ bool better(double fa, int ra, double fb, int rb) { return fa > fb || (fa == fb && (unsigned)ra < (unsigned)rb); }

__device__ __forceinline__ void block_fold(double& far, int& row, float& mo, Part* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double of = __shfl_xor(far, off, 64);
        const int orow = __shfl_xor(row, off, 64);
        const float om = __shfl_xor(mo, off, 64);
        if (better(of, orow, far, row)) { far = of; row = orow; }
        mo = om < mo ? om : mo;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = Part{far, row, mo};
    __syncthreads();
    far = red[0].far; row = red[0].row; mo = red[0].min_other;
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) {
        if (better(red[w].far, red[w].row, far, row)) { far = red[w].far; row = red[w].row; }
        mo = red[w].min_other < mo ? red[w].min_other : mo;
    }
}

__global__ void __launch_bounds__(kBlock) k_dem_step(const float* __restrict__ dist, const int32_t* __restrict__ cls, int n,
                                                     const int32_t* __restrict__ pivots, int ii, double* __restrict__ farsum,
                                                     Part* __restrict__ parts) {
    __shared__ Part red[kBlock / 64];
    const int piv = pivots[ii];
    const int pcls = piv >= 0 ? cls[piv] : 0;
    double best = 0.0;
    int best_row = -1;
    float mo = FLT_MAX;
    for (int j = blockIdx.x * kBlock + threadIdx.x; j < n; j += gridDim.x * kBlock) {   // ascending j per thread
        const float dj = dist[j];
        if (piv >= 0 && cls[j] != pcls && dj < mo) mo = dj;
        const double prev = ii == 0 ? 0.0 : farsum[j];
        const double f = j == piv ? -1000000.0 : prev + (double)dj;
        farsum[j] = f;
        if (f > best) { best = f; best_row = j; }
    }
    block_fold(best, best_row, mo, red);
    if (threadIdx.x == 0) parts[blockIdx.x] = Part{best, best_row, mo};
}

__global__ void __launch_bounds__(kBlock) k_dem_pick(const Part* __restrict__ parts, int nparts, int ii, int n_pivots,
                                                     int32_t* __restrict__ pivots, float* __restrict__ min_other) {
    __shared__ Part red[kBlock / 64];
    double best = 0.0;
    int best_row = -1;
    float mo = FLT_MAX;
    for (int b = threadIdx.x; b < nparts; b += kBlock) {
        const Part p = parts[b];
        if (better(p.far, p.row, best, best_row)) { best = p.far; best_row = p.row; }
        mo = p.min_other < mo ? p.min_other : mo;
    }
    block_fold(best, best_row, mo, red);
    if (threadIdx.x == 0) {
        min_other[ii] = mo;
        if (ii < n_pivots - 1) pivots[ii + 1] = pivots[ii] >= 0 ? best_row : -1;
    }
}

constexpr int kMaxUsed = 32;   // ann.cpp:333-334: only the first 32 pivots are walked at query time
constexpr int kLikBatch = 8;
constexpr int kPinLikRows = 131072;

// lik[q][nu] = sum over the kept pivots i (in order) of (pd[q][i] - table[i][nu])^2, entries with table < 0 skipped (:441).
// pd[q][i] = distance(query q, pivot i). One lane per row; the translation unit is built with -ffp-contract=off.
template <int QB>
__global__ void __launch_bounds__(kBlock) k_dem_lik(const float* __restrict__ table, int n, int used, const float* __restrict__ pd, int nq,
                                                    float* __restrict__ lik) {
    __shared__ float spd[QB][kMaxUsed];
    for (int t = threadIdx.x; t < QB * kMaxUsed; t += kBlock) {
        const int q = t / kMaxUsed, i = t % kMaxUsed;
        spd[q][i] = (q < nq && i < used) ? pd[q * used + i] : 0.0f;
    }
    __syncthreads();
    for (int nu = blockIdx.x * kBlock + threadIdx.x; nu < n; nu += gridDim.x * kBlock) {
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.0f;
        for (int i = 0; i < used; ++i) {
            const float m = table[(size_t)i * n + nu];
            if (m >= 0.0f) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const float tmp = spd[q][i] - m;
                    acc[q] = acc[q] + tmp * tmp;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < QB; ++q)
            if (q < nq) lik[(size_t)q * n + nu] = acc[q];
    }
}

// The rows the reference's index bookkeeping (:431-432) updates a number of times other than once per pivot: they are
// recomputed with their multiplicities. mult[e][i] = how often exception row e is visited by the update loop of pivot i.
__global__ void k_dem_lik_fix(const float* __restrict__ table, int n, int used, const float* __restrict__ pd, int nq, const int32_t* __restrict__ rows,
                              const uint8_t* __restrict__ mult, int nexc, float* __restrict__ lik) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nexc * nq) return;
    const int e = t % nexc, q = t / nexc;
    const int nu = rows[e];
    float acc = 0.0f;
    for (int i = 0; i < used; ++i) {
        const float m = table[(size_t)i * n + nu];
        if (m >= 0.0f) {
            const float tmp = pd[q * used + i] - m;
            for (int r = 0; r < mult[e * kMaxUsed + i]; ++r) acc = acc + tmp * tmp;
        }
    }
    lik[(size_t)q * n + nu] = acc;
}

// out[q][k] = distance(query q, gallery row rows[q][k]) over [start,end): lhs = query (ImageInfo::distance).
// A row of the tiled gallery is one float4 per 1 KiB, so a candidate is a gather. One WAVE per `cpw` candidates: its lanes
// fetch the row's float4s side by side (one memory latency instead of d/4 in a row -- the call is latency, not bandwidth),
// park them in LDS next to the query (read ONCE per workgroup: on small calls it sits in pinned host memory), and lane j
// then runs the reference's loop for candidate j in feature order (fir::accum, un-fused).
// Rows longer than `span` float4s go through LDS in pieces of `span` (the sums carry over in the lanes' registers).
// Dynamic LDS: (1 + 4 * cpw) * span float4.
template <int METRIC>
__global__ void __launch_bounds__(kBlock) k_rows_dist(const float4* __restrict__ gal4, int dp4, int64_t n, const float* __restrict__ queries, int d,
                                                      const int32_t* __restrict__ rows, int m, int start, int end, float* __restrict__ out, int cpw,
                                                      int span) {
    extern __shared__ __attribute__((aligned(16))) float4 rsm[];
    float* qs = (float*)rsm;
    const int q = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float4* mine = rsm + span + (size_t)wave * cpw * span;
    const int k0 = (blockIdx.x * (kBlock / 64) + wave) * cpw;
    const int c0 = start >> 2, c1 = (end - 1) >> 2;
    const int64_t my_row = lane < cpw && k0 + lane < m ? (int64_t)rows[(size_t)q * m + k0 + lane] : -1;
    const bool my_valid = my_row >= 0 && my_row < n;
    float acc = 0.0f;
    for (int cb = c0; cb <= c1; cb += span) {
        const int ce = min(c1, cb + span - 1);                   // chunks [cb, ce] of this piece
        for (int k = cb * 4 + threadIdx.x; k < (ce + 1) * 4; k += kBlock) qs[k - cb * 4] = k < d ? queries[(size_t)q * d + k] : 0.0f;
        for (int j = 0; j < cpw; ++j) {
            const int k = k0 + j;
            const int64_t row = k < m ? (int64_t)rows[(size_t)q * m + k] : -1;
            if (row < 0 || row >= n) continue;                     // wave-uniform
            const float4* __restrict__ base = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
            for (int c = cb + lane; c <= ce; c += 64) mine[(size_t)j * span + (c - cb)] = base[(size_t)c * 64];
        }
        __syncthreads();
        if (my_valid) {
            const float* __restrict__ gv = (const float*)(mine + (size_t)lane * span);
            const int f0 = max(start, cb * 4), f1 = min(end, (ce + 1) * 4);
            for (int f = f0; f < f1; ++f) acc = fir::accum<METRIC>(acc, qs[f - cb * 4], gv[f - cb * 4]);
        }
        __syncthreads();
    }
    if (lane < cpw && k0 + lane < m) out[(size_t)q * m + k0 + lane] = my_valid ? acc / (float)(end - start) : fir::kNotFound;
}

// One thread, queued behind the kernels of a small host-pointer call whose results went to pinned host memory: the call's
// ticket. (A device-wide fence + arrival counter inside the producing kernel costs more than this launch: every
// workgroup's fence is an L2 write-back.)
__global__ void k_dem_ticket(unsigned long long* ticket_word, unsigned long long ticket) {
    __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Last kernel of a small fir_dem_likelihoods call: the pivot distances go to pinned host memory, then the ticket.
__global__ void __launch_bounds__(64) k_dem_publish(const float* __restrict__ pd, int count, float* __restrict__ host_pd,
                                                     unsigned long long* ticket_word, unsigned long long ticket) {
    for (int i = threadIdx.x; i < count; i += 64) host_pd[i] = pd[i];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <typename T> T* as() { return (T*)p; }
};

// The PIVOT build. Table row ii goes to d_table + min(ii, keep_rows) * n (rows past keep_rows share one scratch row), the
// pivot's features to d_pivrows + min(ii, keep_piv) * d. pivots/min_other are left on the device.
int dem_build(fir_gallery* g, const fir_gallery_view& v, int first_pivot, int n_pivots, float* d_table, int keep_rows, float* d_pivrows,
              int keep_piv, int32_t* d_pivots, float* d_min_other) {
    const void* gal4 = nullptr;
    int dp4 = 0;
    if (fir_gallery_tiled_(g, &gal4, &dp4) != FIR_OK || !gal4) return dem_fail(FIR_ERR_STATE, "gallery has no tiled copy");
    const int n = (int)v.n;
    const int nblocks = std::min(kMaxBlocks, (n + kBlock - 1) / kBlock);
    Buf dfar, dparts;
    DEM_HIP(dfar.alloc((size_t)n * 8));
    DEM_HIP(dparts.alloc((size_t)nblocks * sizeof(Part)));
    DEM_HIP(hipMemsetAsync(d_pivots, 0xff, (size_t)n_pivots * 4, v.stream));
    DEM_HIP(hipMemcpyAsync(d_pivots, &first_pivot, 4, hipMemcpyHostToDevice, v.stream));
    for (int ii = 0; ii < n_pivots; ++ii) {
        float* row = d_table + (size_t)std::min(ii, keep_rows) * n;
        float* q = d_pivrows + (size_t)std::min(ii, keep_piv) * v.d;
        hipLaunchKernelGGL(k_dem_gather, dim3(std::max(1, std::min(64, (dp4 + kBlock - 1) / kBlock))), dim3(kBlock), 0, v.stream,
                           (const float4*)gal4, dp4, v.d, d_pivots, ii, q);
        DEM_HIP(hipGetLastError());
        const int rc = fir_range_distances_dev(g, q, 1, 0, v.d, row, v.stream);
        if (rc) return rc;
        hipLaunchKernelGGL(k_dem_step, dim3(nblocks), dim3(kBlock), 0, v.stream, row, v.cls, n, d_pivots, ii, dfar.as<double>(), dparts.as<Part>());
        DEM_HIP(hipGetLastError());
        hipLaunchKernelGGL(k_dem_pick, dim3(1), dim3(kBlock), 0, v.stream, dparts.as<Part>(), nblocks, ii, n_pivots, d_pivots, d_min_other);
        DEM_HIP(hipGetLastError());
    }
    DEM_HIP(hipStreamSynchronize(v.stream));   // dfar/dparts are freed on return
    return FIR_OK;
}

int check_build_args(fir_gallery* g, int32_t first_pivot, int32_t n_pivots, fir_gallery_view* v) {
    if (!g) return dem_fail(FIR_ERR_ARG, "NULL gallery");
    if (fir_gallery_view_(g, v) != FIR_OK) return dem_fail(FIR_ERR_ARG, "bad gallery");
    if (!v->cls) return dem_fail(FIR_ERR_STATE, "gallery was created without class labels");
    if (n_pivots <= 0) return dem_fail(FIR_ERR_ARG, "n_pivots=%d must be positive", n_pivots);
    if (v->n <= 0 || v->n >= (int64_t)1 << 30) return dem_fail(FIR_ERR_ARG, "gallery of %lld rows outside [1, 2^30)", (long long)v->n);
    if (first_pivot < 0 || first_pivot >= v->n) return dem_fail(FIR_ERR_ARG, "first_pivot=%d outside the gallery", first_pivot);
    return FIR_OK;
}

int count_built(const int32_t* pivots, int n_pivots) {
    for (int ii = 0; ii < n_pivots; ++ii)
        if (pivots[ii] < 0) return ii;
    return n_pivots;
}

}  // namespace

struct fir_dem {
    fir_gallery* g = nullptr;        // borrowed
    fir_gallery* pivot_rows = nullptr;   // the kept pivots as a gallery of their own: distance(query, pivot i) is one tiny scan
    fir_gallery_view v;
    int n_pivots = 0, used = 0, built = 0;
    std::vector<int32_t> pivots;
    std::vector<float> min_other;
    std::vector<std::pair<int32_t, int32_t> > order_mods;   // (position, value): where likelihood_indices differs from identity after the pivots
    Buf table, pivrows, exc_rows, exc_mult, q, pd, lik;
    int nexc = 0;
    // galleries up to kPinLikRows rows: queries in and pivot distances / likelihoods out through pinned, device-visible host
    // memory the kernels address directly, completion by a ticket word (no copy engine, no stream synchronisation)
    void* pin = nullptr;
    unsigned long long ticket = 0;
};

extern "C" {

int fir_dem_pivot_table(fir_gallery* g, int32_t first_pivot, int32_t n_pivots, int32_t* pivots_out, float* table_out, float* min_other_out,
                        int32_t* n_built_out) {
    fir_gallery_view v;
    int rc = check_build_args(g, first_pivot, n_pivots, &v);
    if (rc) return rc;
    if (!pivots_out) return dem_fail(FIR_ERR_ARG, "pivots_out is NULL");
    DEM_HIP(hipSetDevice(v.device));
    const int keep = table_out ? n_pivots - 1 : 0;
    Buf dtable, dq, dpiv, dmo;
    DEM_HIP(dtable.alloc((size_t)(keep + 1) * v.n * 4));
    DEM_HIP(dq.alloc((size_t)v.d * 4));
    DEM_HIP(dpiv.alloc((size_t)n_pivots * 4));
    DEM_HIP(dmo.alloc((size_t)n_pivots * 4));
    if ((rc = dem_build(g, v, first_pivot, n_pivots, dtable.as<float>(), keep, dq.as<float>(), 0, dpiv.as<int32_t>(), dmo.as<float>()))) return rc;
    DEM_HIP(hipMemcpy(pivots_out, dpiv.p, (size_t)n_pivots * 4, hipMemcpyDeviceToHost));
    if (min_other_out) DEM_HIP(hipMemcpy(min_other_out, dmo.p, (size_t)n_pivots * 4, hipMemcpyDeviceToHost));
    if (table_out) DEM_HIP(hipMemcpy(table_out, dtable.p, (size_t)n_pivots * v.n * 4, hipMemcpyDeviceToHost));
    if (n_built_out) *n_built_out = count_built(pivots_out, n_pivots);
    return FIR_OK;
}

int fir_dem_create(fir_gallery* g, int32_t first_pivot, int32_t n_pivots, fir_dem** out) {
    fir_gallery_view v;
    int rc = check_build_args(g, first_pivot, n_pivots, &v);
    if (rc) return rc;
    if (!out) return dem_fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    int32_t metric = 0;
    if ((rc = fir_gallery_info(g, nullptr, nullptr, &metric, nullptr))) return rc;
    DEM_HIP(hipSetDevice(v.device));
    fir_dem* h = new fir_dem();
    struct Guard { fir_dem* h; ~Guard() { if (h) fir_dem_destroy(h); } } guard{h};
    h->g = g; h->v = v; h->n_pivots = n_pivots;
    const int keep = std::min<int>(n_pivots, kMaxUsed);
    const int n = (int)v.n;
    Buf dpiv, dmo;
    DEM_HIP(h->table.alloc((size_t)(keep + 1) * n * 4));
    DEM_HIP(h->pivrows.alloc((size_t)(keep + 1) * v.d * 4));
    DEM_HIP(dpiv.alloc((size_t)n_pivots * 4));
    DEM_HIP(dmo.alloc((size_t)n_pivots * 4));
    if ((rc = dem_build(g, v, first_pivot, n_pivots, h->table.as<float>(), keep, h->pivrows.as<float>(), keep, dpiv.as<int32_t>(), dmo.as<float>())))
        return rc;
    h->pivots.resize((size_t)n_pivots);
    h->min_other.resize((size_t)n_pivots);
    DEM_HIP(hipMemcpy(h->pivots.data(), dpiv.p, (size_t)n_pivots * 4, hipMemcpyDeviceToHost));
    DEM_HIP(hipMemcpy(h->min_other.data(), dmo.p, (size_t)n_pivots * 4, hipMemcpyDeviceToHost));
    h->built = count_built(h->pivots.data(), n_pivots);
    h->used = std::min(h->built, kMaxUsed);
    if ((rc = fir_gallery_create_dev(h->pivrows.as<float>(), h->used, v.d, nullptr, metric, v.device, v.stream, &h->pivot_rows))) return rc;

    // The reference keeps the candidates in an index array and moves each pivot to its front with two plain writes
    // (ann.cpp:431-432); the update loop (:437-446) then runs over the POSITIONS behind the front. Replay that on a
    // sparse copy: which rows does each pivot's loop visit, and how often? Only rows in {0..used-1} + {pivots} can
    // deviate from "once per pivot".
    std::unordered_map<int32_t, int32_t> mod;   // position -> value, where it is not the identity
    auto at = [&](int32_t pos) { auto it = mod.find(pos); return it == mod.end() ? pos : it->second; };
    std::vector<int32_t> special;
    for (int i = 0; i < h->used; ++i) special.push_back(i);
    for (int i = 0; i < h->used; ++i) special.push_back(h->pivots[(size_t)i]);
    std::sort(special.begin(), special.end());
    special.erase(std::unique(special.begin(), special.end()), special.end());
    std::vector<uint8_t> mult(special.size() * kMaxUsed, 0);
    for (int i = 0; i < h->used; ++i) {
        const int32_t p = h->pivots[(size_t)i];
        mod[p] = at(i);
        mod[i] = p;
        for (size_t e = 0; e < special.size(); ++e) {
            const int32_t w = special[e];
            int cnt = (w > i && mod.find(w) == mod.end()) ? 1 : 0;
            for (const auto& kv : mod)
                if (kv.first > i && kv.second == w) ++cnt;
            mult[e * kMaxUsed + i] = (uint8_t)std::min(cnt, 255);
        }
    }
    for (const auto& kv : mod)
        if (kv.first != kv.second) h->order_mods.push_back(kv);
    std::sort(h->order_mods.begin(), h->order_mods.end());
    h->nexc = (int)special.size();
    DEM_HIP(h->exc_rows.alloc(special.size() * 4));
    DEM_HIP(h->exc_mult.alloc(mult.size()));
    DEM_HIP(hipMemcpy(h->exc_rows.p, special.data(), special.size() * 4, hipMemcpyHostToDevice));
    DEM_HIP(hipMemcpy(h->exc_mult.p, mult.data(), mult.size(), hipMemcpyHostToDevice));
    DEM_HIP(h->q.alloc((size_t)kLikBatch * v.d * 4));
    DEM_HIP(h->pd.alloc((size_t)kLikBatch * kMaxUsed * 4));
    DEM_HIP(h->lik.alloc((size_t)kLikBatch * n * 4));
    if (n <= kPinLikRows) {
        const size_t bytes = (size_t)kLikBatch * ((size_t)v.d + kMaxUsed + (size_t)n) * 4 + 64;
        DEM_HIP(hipHostMalloc(&h->pin, bytes, hipHostMallocDefault));
        std::memset(h->pin, 0, bytes);
    }
    guard.h = nullptr;
    *out = h;
    return FIR_OK;
}

int fir_dem_destroy(fir_dem* h) {
    if (!h) return FIR_OK;
    (void)hipSetDevice(h->v.device);
    if (h->pivot_rows) fir_gallery_destroy(h->pivot_rows);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return FIR_OK;
}

int fir_dem_info(const fir_dem* h, int32_t* n_pivots, int32_t* n_built, int32_t* n_used, int64_t* n) {
    if (!h) return dem_fail(FIR_ERR_ARG, "NULL handle");
    if (n_pivots) *n_pivots = h->n_pivots;
    if (n_built) *n_built = h->built;
    if (n_used) *n_used = h->used;
    if (n) *n = h->v.n;
    return FIR_OK;
}

int fir_dem_get(fir_dem* h, int32_t* pivots_out, float* min_other_out, float* table_out, int32_t* order_out) {
    if (!h) return dem_fail(FIR_ERR_ARG, "NULL handle");
    if (pivots_out) std::copy(h->pivots.begin(), h->pivots.end(), pivots_out);
    if (min_other_out) std::copy(h->min_other.begin(), h->min_other.end(), min_other_out);
    if (table_out) {
        DEM_HIP(hipSetDevice(h->v.device));
        DEM_HIP(hipMemcpy(table_out, h->table.p, (size_t)h->used * h->v.n * 4, hipMemcpyDeviceToHost));
    }
    if (order_out) {
        for (int64_t i = 0; i < h->v.n; ++i) order_out[i] = (int32_t)i;
        for (const auto& kv : h->order_mods) order_out[kv.first] = kv.second;
    }
    return FIR_OK;
}

int fir_dem_likelihoods(fir_dem* h, const float* queries, int32_t qb, float* pivot_dist_out, float* lik_out) {
    if (!h || (qb > 0 && !queries)) return dem_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return dem_fail(FIR_ERR_ARG, "qb < 0");
    const fir_gallery_view& v = h->v;
    DEM_HIP(hipSetDevice(v.device));
    const int n = (int)v.n, used = h->used;
    const int nblocks = std::min(kMaxBlocks, (n + kBlock - 1) / kBlock);
    float* hq = (float*)h->pin;
    float* hpd = hq ? hq + (size_t)kLikBatch * v.d : nullptr;
    float* hlik = hq ? hpd + (size_t)kLikBatch * kMaxUsed : nullptr;
    unsigned long long* tword = hq ? (unsigned long long*)(((uintptr_t)(hlik + (size_t)kLikBatch * n) + 7) & ~(uintptr_t)7) : nullptr;
    for (int q0 = 0; q0 < qb; q0 += kLikBatch) {
        const int nq = std::min(kLikBatch, qb - q0);
        const float* dq = h->q.as<float>();
        float* dlik = h->lik.as<float>();
        if (hq) {
            std::memcpy(hq, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4);
            dq = hq;
            dlik = hlik;
        } else {
            DEM_HIP(hipMemcpyAsync(h->q.p, queries + (size_t)q0 * v.d, (size_t)nq * v.d * 4, hipMemcpyHostToDevice, v.stream));
        }
        const int rc = fir_range_distances_dev(h->pivot_rows, dq, nq, 0, v.d, h->pd.as<float>(), v.stream);   // pd[q][used]
        if (rc) return rc;
        if (!hq && pivot_dist_out)
            DEM_HIP(hipMemcpyAsync(pivot_dist_out + (size_t)q0 * used, h->pd.p, (size_t)nq * used * 4, hipMemcpyDeviceToHost, v.stream));
        if (lik_out) {
            hipLaunchKernelGGL(k_dem_lik<kLikBatch>, dim3(nblocks), dim3(kBlock), 0, v.stream, h->table.as<float>(), n, used, h->pd.as<float>(), nq, dlik);
            DEM_HIP(hipGetLastError());
            hipLaunchKernelGGL(k_dem_lik_fix, dim3((h->nexc * nq + 63) / 64), dim3(64), 0, v.stream, h->table.as<float>(), n, used, h->pd.as<float>(),
                               nq, h->exc_rows.as<int32_t>(), h->exc_mult.as<uint8_t>(), h->nexc, dlik);
            DEM_HIP(hipGetLastError());
            if (!hq) DEM_HIP(hipMemcpyAsync(lik_out + (size_t)q0 * n, h->lik.p, (size_t)nq * n * 4, hipMemcpyDeviceToHost, v.stream));
        }
        if (!hq) {
            DEM_HIP(hipStreamSynchronize(v.stream));
            continue;
        }
        const unsigned long long ticket = ++h->ticket;
        hipLaunchKernelGGL(k_dem_publish, dim3(1), dim3(64), 0, v.stream, h->pd.as<float>(), nq * used, hpd, tword, ticket);
        DEM_HIP(hipGetLastError());
        const auto t0 = std::chrono::steady_clock::now();
        for (int spins = 0; __atomic_load_n(tword, __ATOMIC_ACQUIRE) != ticket; ++spins) {
            if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                DEM_HIP(hipStreamSynchronize(v.stream));
                if (__atomic_load_n(tword, __ATOMIC_ACQUIRE) != ticket) return dem_fail(FIR_ERR_HIP, "the result ticket was not published");
                break;
            }
        }
        if (pivot_dist_out) std::memcpy(pivot_dist_out + (size_t)q0 * used, hpd, (size_t)nq * used * 4);
        if (lik_out) std::memcpy(lik_out + (size_t)q0 * n, hlik, (size_t)nq * n * 4);
    }
    return FIR_OK;
}

int fir_rows_distances(fir_gallery* g, const float* queries, int32_t qb, const int32_t* rows, int32_t m, int32_t start_pos, int32_t end_pos,
                       float* out) {
    fir_gallery_view v;
    if (!g || (qb > 0 && m > 0 && (!queries || !rows || !out))) return dem_fail(FIR_ERR_ARG, "NULL argument");
    if (fir_gallery_view_(g, &v) != FIR_OK) return dem_fail(FIR_ERR_ARG, "bad gallery");
    if (qb < 0 || m < 0 || qb > 65535) return dem_fail(FIR_ERR_ARG, "qb=%d / m=%d out of range", qb, m);
    if (end_pos == 0) end_pos = v.d;
    if (start_pos < 0 || end_pos > v.d || start_pos >= end_pos) return dem_fail(FIR_ERR_ARG, "feature range [%d,%d) outside [0,%d)", start_pos, end_pos, v.d);
    if (qb == 0 || m == 0) return FIR_OK;
    const void* gal4 = nullptr;
    int dp4 = 0;
    if (fir_gallery_tiled_(g, &gal4, &dp4) != FIR_OK || !gal4) return dem_fail(FIR_ERR_STATE, "gallery has no tiled copy");
    int32_t metric = 0;
    int rc = fir_gallery_info(g, nullptr, nullptr, &metric, nullptr);
    if (rc) return rc;
    DEM_HIP(hipSetDevice(v.device));
    // candidates per wave: 4 while the workgroup's 1 + 16 rows fit 64 KiB of LDS (d <= 960), else 1; rows beyond 2 048 features
    // go through LDS in pieces of 512 float4
    const int cpw = (size_t)17 * dp4 * 16 <= 64 * 1024 ? 4 : 1;
    const int span = std::min(dp4, 512);
    const size_t lds = (size_t)(1 + (kBlock / 64) * cpw) * span * 16;
    const int per_block = (kBlock / 64) * cpw;
    const dim3 grid((m + per_block - 1) / per_block, qb);
#define FIR_ROWS_LAUNCH(M, Q, R, O)                                                                                                       \
    hipLaunchKernelGGL(k_rows_dist<M>, grid, dim3(kBlock), lds, v.stream, (const float4*)gal4, dp4, v.n, Q, v.d, R, m, start_pos, end_pos, O, cpw, span)
#define FIR_ROWS_BY_METRIC(Q, R, O)                                                                                                       \
    do {                                                                                                                                  \
        if (metric == FIR_METRIC_L2) FIR_ROWS_LAUNCH(fir::kL2, Q, R, O);                                                                  \
        else if (metric == FIR_METRIC_CHI2) FIR_ROWS_LAUNCH(fir::kChi2, Q, R, O);                                                         \
        else FIR_ROWS_LAUNCH(fir::kKL, Q, R, O);                                                                                          \
    } while (0)
    const size_t qbytes = ((size_t)qb * v.d * 4 + 15) & ~(size_t)15, rbytes = (size_t)qb * m * 4;
    // Small calls (the DEM walk: one query, a few hundred candidate rows): everything through the handle's pinned,
    // device-visible buffer -- no allocation, no copy engine, no stream synchronisation (a ticket written behind the kernel).
    void* pin_base = nullptr;
    size_t pin_cap = 0;
    uint64_t* pin_res = nullptr;
    if ((size_t)qb * m <= 8000 && fir_gallery_pin_(g, &pin_base, &pin_cap, &pin_res) == FIR_OK && qbytes + rbytes <= pin_cap) {
        float* hq = (float*)pin_base;
        int32_t* hr = (int32_t*)((char*)pin_base + qbytes);
        float* ho = (float*)pin_res;
        unsigned long long* tword = (unsigned long long*)(pin_res + 4095);
        std::memcpy(hq, queries, (size_t)qb * v.d * 4);
        std::memcpy(hr, rows, rbytes);
        const unsigned long long ticket = fir_gallery_next_ticket_(g);
        FIR_ROWS_BY_METRIC(hq, hr, ho);
        hipLaunchKernelGGL(k_dem_ticket, dim3(1), dim3(1), 0, v.stream, tword, ticket);
        const hipError_t le = hipGetLastError();
        if (le == hipSuccess && fir_gallery_wait_ticket_(g, (volatile uint64_t*)tword, ticket) == FIR_OK) {
            std::memcpy(out, ho, rbytes);
            return FIR_OK;
        }
        (void)hipStreamSynchronize(v.stream);   // the launch failed or never published: the general path below reports why
    }
    void *dq = nullptr, *drows = nullptr, *dout = nullptr;
    if ((rc = fir_gallery_scratch_(g, 8, (size_t)qb * v.d * 4, &dq))) return rc;
    if ((rc = fir_gallery_scratch_(g, 9, rbytes, &drows))) return rc;
    if ((rc = fir_gallery_scratch_(g, 10, rbytes, &dout))) return rc;
    DEM_HIP(hipMemcpyAsync(dq, queries, (size_t)qb * v.d * 4, hipMemcpyHostToDevice, v.stream));
    DEM_HIP(hipMemcpyAsync(drows, rows, rbytes, hipMemcpyHostToDevice, v.stream));
    FIR_ROWS_BY_METRIC((const float*)dq, (const int32_t*)drows, (float*)dout);
#undef FIR_ROWS_BY_METRIC
#undef FIR_ROWS_LAUNCH
    DEM_HIP(hipGetLastError());
    DEM_HIP(hipMemcpyAsync(out, dout, rbytes, hipMemcpyDeviceToHost, v.stream));
    DEM_HIP(hipStreamSynchronize(v.stream));
    return FIR_OK;
}

}  // extern "C"
