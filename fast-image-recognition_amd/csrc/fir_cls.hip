// fir_cls.hip -- double-precision kNN / PNN classifiers of qt_cpp/classification.cpp on gfx950.
//
// Same design as the f32 matcher (fir_kernels.h): the training rows are re-tiled once into
//   tile t (64 rows) x chunk c (2 features) x lane r  ->  double2 at gal2[(t*dp2 + c)*64 + r]
// holding (g - avg) -- Classifier::normalize of the training side (classification.cpp:103-105,
// 132) evaluated once instead of per query: same double subtraction, same bits. A wave owns a
// tile, a lane owns a row and accumulates  diff = (g-avg) - (q-avg);  dist += diff*diff  in
// feature order with un-fused mul/add (translation unit built -ffp-contract=off), i.e. the
// reference's evaluation order (classification.cpp:123-141, 199-211): the distance sums are
// bit-identical to the reference's doubles. exp() and the per-class sums of PNN are a parallel
// reduction (device exp, different summation order): scores agree to ~1e-13 relative, the
// arg-max class is what is compared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kTileRows = 64;
constexpr int kQB = 4;       // queries per pass of the f64 scan (training sets that stay in the caches, the sequential PNN)
constexpr int kQBBig = 8;    // ... of a training set streamed from HBM: the pass is HBM-bound up to 8 queries (3 f64 vector ops per element and query)
constexpr int kKMax = 8;

int cls_fail(int code, const char* fmt, ...);   // defined with the C ABI below

typedef const double __attribute__((address_space(4)))* sdouble_p;

// rows[slab][d] row-major -> tiled (g - avg). One thread per output double2.
__global__ void __launch_bounds__(kBlock) k_cls_retile(const double* __restrict__ rows, int64_t slab_rows, int64_t row0, int64_t nt,
                                                        int d, int dp2, const double* __restrict__ avg, double2* __restrict__ gal2) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t slab_tiles = (slab_rows + kTileRows - 1) / kTileRows;
    if (o >= slab_tiles * dp2 * 64) return;
    const int r = (int)(o & 63);
    const int64_t tc = o >> 6;
    const int c = (int)(tc % dp2);
    const int64_t tl = tc / dp2;
    const int64_t lrow = tl * kTileRows + r, grow = row0 + lrow;
    double v[2] = {0.0, 0.0};
    if (lrow < slab_rows && grow < nt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = c * 2 + j;
            if (k < d) v[j] = rows[lrow * d + k] - avg[k];          // normalize(), classification.cpp:103-105
        }
    }
    gal2[((row0 / kTileRows + tl) * dp2 + c) * 64 + r] = make_double2(v[0], v[1]);
}

// queries[nq][d] -> qn[k][QB] = q[k] - avg[k] (the query side of normalize(), :135), zero padded.
template <int QB>
__global__ void __launch_bounds__(kBlock) k_cls_prep_queries(const double* __restrict__ q, int nq, int d, int dp2,
                                                              const double* __restrict__ avg, double* __restrict__ qn) {
    const int kk = dp2 * 2;
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= (int64_t)kk * QB) return;
    const int qi = (int)(o % QB);
    const int k = (int)(o / QB);
    qn[o] = (qi < nq && k < d) ? q[(int64_t)qi * d + k] - avg[k] : 0.0;
}

// sums[q][row] = sum_k ((g-avg) - (q-avg))^2, sequential in k.
// lo, hi: double2-chunk range [lo, hi) of the features (whole scan: 0, dp2). cstep < hi - lo: the range is cut into
// consecutive sub-ranges of cstep chunks, each a fresh sum written `sub_stride` doubles after the previous one (the
// 32-feature chunks of the sequential PNN, classification.cpp:245-262, from ONE pass over the training rows).
template <int QB>
__global__ void __launch_bounds__(kBlock) k_cls_scan(const double2* __restrict__ gal2, const double* __restrict__ qn, int64_t nt,
                                                      int tiles, int dp2, int d, int waves, int nq, int lo, int hi,
                                                      double* __restrict__ sums_base, int cstep, int64_t sub_stride) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    sdouble_p qc = (sdouble_p)(uintptr_t)qn;
    for (int t = gw; t < tiles; t += waves) {
      const double2* p = gal2 + (size_t)t * dp2 * 64 + lane;
      double* sums = sums_base;
      for (int c0 = lo; c0 < hi; c0 += cstep, sums += sub_stride) {
        const int c1 = min(hi, c0 + cstep);
        double acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.0;
        int c = c0;
        for (; c + 4 <= c1; c += 4) {
            double2 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = p[(size_t)(c + u) * 64];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double gv[2] = {g[u].x, g[u].y};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = (c + u) * 2 + j;
                    if (k < d) {
#pragma unroll
                        for (int q = 0; q < QB; ++q) {
                            const double diff = gv[j] - qc[k * QB + q];       // :132-137
                            acc[q] = acc[q] + diff * diff;                      // :141
                        }
                    }
                }
            }
        }
        for (; c < c1; ++c) {
            const double2 g = p[(size_t)c * 64];
            const double gv[2] = {g.x, g.y};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int k = c * 2 + j;
                if (k < d) {
#pragma unroll
                    for (int q = 0; q < QB; ++q) {
                        const double diff = gv[j] - qc[k * QB + q];
                        acc[q] = acc[q] + diff * diff;
                    }
                }
            }
        }
        const int64_t row = (int64_t)t * kTileRows + lane;
        if (row < nt) {
#pragma unroll
            for (int q = 0; q < QB; ++q)
                if (q < nq) sums[(size_t)q * nt + row] = acc[q];
        }
      }
    }
}

// ---- the whole-range scan of a training set streamed from HBM (K3 at scale: 1M x 512 doubles = 4.1 GB per pass) ----
// k_cls_scan above is the compiler-scheduled form: four double2 loads in flight per lane and the query values through the
// scalar cache -- eight queries x 512 features x 8 B = 32 KiB per tile against a 16 KiB scalar cache that every wave of the
// CU walks at its own pace: the same thrash profiles/r01_sweep_notes.md found in k_scan_l2_fast. At eight queries per pass it
// reached 47 % of the HBM rate with the f64 pipes a third busy. This is k_scan_l2_lds's structure in double precision:
//   * the query tile (8 queries x dp2 * 2 features, q - avg) sits in LDS, [feature][query]; a feature's eight values are four
//     ds_read_b128 at one address for all lanes (a broadcast), read one feature ahead of the arithmetic;
//   * the training rows stream tile -> VGPR with non-temporal 16-byte loads, a register double buffer of U double2 per lane
//     (U = 8: 8 KiB per wave in flight behind the group being consumed);
//   * per (feature, query): v_add_f64 (the subtraction), v_mul_f64, v_add_f64 -- un-fused, in feature order, exactly the
//     reference's evaluation order (classification.cpp:132-141): the sums are the same bits as k_cls_scan's and the oracle's;
//   * all query tiles of a call go into ONE launch (blockIdx.y = tile of eight queries): consecutive passes run back to back
//     on the chip instead of draining at a launch boundary each.
// Padding features (k >= d) are +0 on both sides: diff * diff = +0 added to a non-negative sum leaves its bits alone.
// Dynamic LDS: (dp2 * 2 + 1) * 8 doubles.
__device__ __forceinline__ double2 cls_ld_nt(const double2* p) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d v = __builtin_nontemporal_load((const v2d*)p);
    return make_double2(v.x, v.y);
}
// NT = tiles of eight queries per read of the training rows (blockIdx.y = group of NT consecutive tiles): at 8 queries the pass is
// 0.72 of HBM with the f64 vector pipes about half busy (24 un-fused f64 operations per feature and tile); with two tiles per read
// the vector pipes become the bound and a query costs about half the time. Every load group is used for the tiles in turn, each
// with its own sums; per (row, query) the operations and their order are unchanged.
// BLOCK = threads per workgroup.
// R = tiles of 64 training rows per wave and step (a lane owns R rows): every LDS read of a query value then serves R rows. With one row
// per lane a feature of one query tile is four broadcast ds_read_b128 against 24 f64 operations -- at two tiles per read the CU's LDS
// pipe is two thirds busy when the vector pipes are full, and with only two waves per SIMD the reads' latency under that load shows
// (profiles/r04_k3_*.txt); two rows per lane halve the LDS traffic per operation and double the work behind every read.
// WPS = waves per SIMD the launch bounds ask for (registers: 512 / WPS per lane).
template <int kClsU, int NT, int BLOCK = kBlock, int R = 1, int WPS = (NT > 1 ? 2 : 4)>
__global__ void __launch_bounds__(BLOCK, WPS) k_cls_scan_lds(const double2* __restrict__ gal2, const double* __restrict__ qn_tiles, int64_t nt,
                                                                            int tiles, int dp2, int waves, int nq_total, double* __restrict__ sums_base) {
    extern __shared__ __attribute__((aligned(16))) double2 lqd[];          // per tile: [(feature k) * 4 + i] = queries 2i, 2i + 1 of feature k
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
    const int kk = dp2 * 2;
    const int n2 = kk * 4;                                                 // double2 per tile (+ 4 of zeroed slack)
    const int tile0 = (int)blockIdx.y * NT;
    const int ntiles_total = (nq_total + 7) / 8;
    {
#pragma unroll
        for (int h = 0; h < NT; ++h) {
            const bool live = tile0 + h < ntiles_total;
            const double2* src = (const double2*)(qn_tiles + (size_t)(tile0 + (live ? h : 0)) * kk * 8);
            for (int i = threadIdx.x; i < n2 + 4; i += BLOCK) lqd[(size_t)h * (n2 + 4) + i] = i < n2 ? src[i] : make_double2(0.0, 0.0);
        }
        __syncthreads();
    }
    // (as k_scan_l2_lds: an LDS base the compiler cannot prove uniform keeps the reads on one address register + immediate offsets)
    int zero_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const double2* lq0 = lqd + zero_v;
    const int tsteps = (tiles + R - 1) / R;                                // a wave's step: R consecutive tiles (the last one may repeat the last tile)
    for (int ts = gw; ts < tsteps; ts += waves) {
        const double2* p[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = ts * R + r < tiles ? ts * R + r : tiles - 1;
            p[r] = gal2 + (size_t)t * dp2 * 64 + lane;
        }
        double acc[R][NT][8];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int h = 0; h < NT; ++h)
#pragma unroll
                for (int q = 0; q < 8; ++q) acc[r][h][q] = 0.0;
        // COUNT chunks (two features each) from chunk c on of this lane's R rows against one tile's eight queries; the LDS reads run one feature
        // ahead (`cur` = the tile's query values of the NEXT feature to be used: carried from call to call, so that switching between the tiles
        // of a read does not restart the read-ahead). Per (row, query) the operations and their order are the reference's whatever R and NT are.
        auto run = [&](int h, double2 (&cur)[4], const double2* lq, const double2 (&g)[R][kClsU], int c, auto count_tag) {
            constexpr int COUNT = decltype(count_tag)::value;
#pragma unroll
            for (int u = 0; u < COUNT; ++u) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    double2 nx[4];
                    const int k1 = (c + u) * 2 + j + 1;                             // (past the last feature: the zeroed slack)
#pragma unroll
                    for (int i = 0; i < 4; ++i) nx[i] = lq[k1 * 4 + i];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const double gv = j == 0 ? g[r][u].x : g[r][u].y;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const double d0 = gv - cur[i].x;                        // classification.cpp:132-137
                            const double d1 = gv - cur[i].y;
                            acc[r][h][2 * i] = acc[r][h][2 * i] + d0 * d0;          // :141
                            acc[r][h][2 * i + 1] = acc[r][h][2 * i + 1] + d1 * d1;
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < 4; ++i) cur[i] = nx[i];
                }
            }
        };
        int c = 0;
        const int ng = dp2 / kClsU;
        double2 g[R][kClsU];
        if (ng > 0) {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < kClsU; ++u) g[r][u] = cls_ld_nt(p[r] + (size_t)u * 64);
        }
        double2 curq[NT][4];
#pragma unroll
        for (int h = 0; h < NT; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) curq[h][i] = lq0[(size_t)h * (n2 + 4) + i];
        for (int gi = 0; gi < ng; ++gi, c += kClsU) {
            double2 nxg[R][kClsU];
            const int gn = gi + 1 < ng ? gi + 1 : gi;                           // the last group re-reads itself (keeps the loop branch-free)
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < kClsU; ++u) nxg[r][u] = cls_ld_nt(p[r] + (size_t)(gn * kClsU + u) * 64);
#pragma unroll
            for (int h = 0; h < NT; ++h) run(h, curq[h], lq0 + (size_t)h * (n2 + 4), g, c, std::integral_constant<int, kClsU>());
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int u = 0; u < kClsU; ++u) g[r][u] = nxg[r][u];
        }
        for (; c < dp2; ++c) {
            double2 gt[R][kClsU];
#pragma unroll
            for (int r = 0; r < R; ++r) gt[r][0] = cls_ld_nt(p[r] + (size_t)c * 64);
#pragma unroll
            for (int h = 0; h < NT; ++h) run(h, curq[h], lq0 + (size_t)h * (n2 + 4), gt, c, std::integral_constant<int, 1>());
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = ts * R + r;
            const int64_t row = (int64_t)t * kTileRows + lane;
            if (t < tiles && row < nt) {
#pragma unroll
                for (int h = 0; h < NT; ++h) {
                    const int q0 = (tile0 + h) * 8;
                    const int nq = nq_total - q0 < 8 ? nq_total - q0 : 8;
                    double* sums = sums_base + (size_t)q0 * nt;
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (q < nq) sums[(size_t)q * nt + row] = acc[r][h][q];
                }
            }
        }
    }
}

// queries[nq][d] -> tiles of eight: qn[tile][k][8] = q[k] - avg[k], zero padded (the query side of normalize(), :135)
__global__ void __launch_bounds__(kBlock) k_cls_prep_query_tiles(const double* __restrict__ q, int nq, int d, int dp2, const double* __restrict__ avg,
                                                                  double* __restrict__ qn) {
    const int kk = dp2 * 2;
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t ntile = (nq + 7) / 8;
    if (o >= ntile * kk * 8) return;
    const int qi8 = (int)(o & 7);
    const int k = (int)((o >> 3) % kk);
    const int64_t tile = (o >> 3) / kk;
    const int64_t qi = tile * 8 + qi8;
    qn[o] = (qi < nq && k < d) ? q[qi * d + k] - avg[k] : 0.0;
}

// The same sums for ONE query (the reference's predict() per test vector, ~3 000 x 256 training rows): such a call is
// latency, not bandwidth -- 48 waves, each walking its 128 double2 loads four at a time, took 52 us. Here every wave keeps
// 16 loads in flight, and the query side of normalize() (q - avg, :135) is computed by the workgroup itself into LDS, which
// saves the separate k_cls_prep_queries launch. Same operations per element, same order in k.
// Dynamic LDS: 2 * dp2 doubles.
__global__ void __launch_bounds__(kBlock) k_cls_scan_one(const double2* __restrict__ gal2, const double* __restrict__ q,
                                                          const double* __restrict__ avg, int64_t nt, int tiles, int dp2, int d, int waves, int lo,
                                                          int hi, double* __restrict__ sums_base, int cstep, int64_t sub_stride) {
    extern __shared__ __attribute__((aligned(16))) double qs1[];
    for (int k = threadIdx.x; k < dp2 * 2; k += kBlock) qs1[k] = k < d ? q[k] - avg[k] : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    constexpr int U = 16;
    for (int t = gw; t < tiles; t += waves) {
        const double2* p = gal2 + (size_t)t * dp2 * 64 + lane;
        double* sums = sums_base;
        for (int c0 = lo; c0 < hi; c0 += cstep, sums += sub_stride) {
            const int c1 = min(hi, c0 + cstep);
            double acc = 0.0;
            int c = c0;
            for (; c + U <= c1; c += U) {
                double2 g[U];
#pragma unroll
                for (int u = 0; u < U; ++u) g[u] = p[(size_t)(c + u) * 64];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const double gv[2] = {g[u].x, g[u].y};
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int k = (c + u) * 2 + j;
                        if (k < d) {
                            const double diff = gv[j] - qs1[k];             // :132-137
                            acc = acc + diff * diff;                        // :141
                        }
                    }
                }
            }
            for (; c < c1; ++c) {
                const double2 g = p[(size_t)c * 64];
                const double gv[2] = {g.x, g.y};
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int k = c * 2 + j;
                    if (k < d) {
                        const double diff = gv[j] - qs1[k];
                        acc = acc + diff * diff;
                    }
                }
            }
            const int64_t row = (int64_t)t * kTileRows + lane;
            if (row < nt) sums[row] = acc;
        }
    }
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// PNN class outputs, classification.cpp:195-216: scores[q][c] = sum_{t in class c} exp(-dist/(2 d var)) / nt.
// grid (num_classes, qb), one wave per (class, query).
__global__ void __launch_bounds__(64) k_cls_pnn(const double* __restrict__ sums, const int32_t* __restrict__ class_off, int64_t nt,
                                                 int num_classes, double denom /* 2*d*var */, double den /* total_training_size */,
                                                 double* __restrict__ scores) {
    const int c = blockIdx.x, q = blockIdx.y;
    const double* s = sums + (size_t)q * nt;
    double acc = 0.0;
    for (int t = class_off[c] + threadIdx.x; t < class_off[c + 1]; t += 64) acc += exp(-s[t] / denom);
    acc = wave_sum(acc);
    if (threadIdx.x == 0) scores[(size_t)q * num_classes + c] = acc / den;
}

// kNN: the class that first collects k votes in the globally sorted order (classification.cpp:151-160)
// is the class whose k-th nearest member is nearest. kth[q][c] = k-th smallest mean distance of
// class c (+inf when the class has fewer than k rows). One wave per (class, query).
__global__ void __launch_bounds__(64) k_cls_knn_kth(const double* __restrict__ sums, const int32_t* __restrict__ class_off, int64_t nt,
                                                     int num_classes, int d, int k, double* __restrict__ kth,
                                                     double* __restrict__ nearest /* NULL, or [q][class][k]: the k smallest, ascending */) {
    const int c = blockIdx.x, q = blockIdx.y;
    const double* s = sums + (size_t)q * nt;
    double best[kKMax];
#pragma unroll
    for (int i = 0; i < kKMax; ++i) best[i] = DBL_MAX;
    for (int t = class_off[c] + threadIdx.x; t < class_off[c + 1]; t += 64) {
        double v = s[t] / (double)d;                                              // :143
#pragma unroll
        for (int i = 0; i < kKMax; ++i) {
            const bool sw = v < best[i];
            const double tmp = best[i];
            best[i] = sw ? v : tmp;
            v = sw ? tmp : v;
        }
    }
    // k rounds of wave-min with removal (each lane pops its own head when it wins; ties pop one lane)
    double res = DBL_MAX;
    for (int r = 0; r < k; ++r) {
        double m = best[0];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(m, off, 64);
            m = o < m ? o : m;
        }
        res = m;
        if (nearest && threadIdx.x == 0) nearest[((size_t)q * num_classes + c) * k + r] = m;   // DBL_MAX once the class is exhausted
        const unsigned long long who = __ballot(best[0] == m);
        const int winner = __ffsll((long long)who) - 1;
        if ((int)threadIdx.x == winner) {
#pragma unroll
            for (int i = 0; i + 1 < kKMax; ++i) best[i] = best[i + 1];
            best[kKMax - 1] = DBL_MAX;
        }
    }
    if (threadIdx.x == 0) kth[(size_t)q * num_classes + c] = (class_off[c + 1] - class_off[c] >= k) ? res : DBL_MAX;
}

// PNNClassifier::predict_sequentional, classification.cpp:228-295. cs[chunk][q][nt]: per-row sums of
// 32-feature chunk `chunk`; dist[q][nt] workspace. One workgroup per query; wave w owns classes
// w, w+4, ... (rows of a class are contiguous), so every class output is summed in a fixed order.
// Dynamic LDS: num_classes doubles (outputs) + num_classes ints (classes_to_check) + num_classes + 1 ints (class offsets).
// 1024 threads per query. Every chunk is three short phases instead of one wave walking its classes one after the other
// (25 classes x 8 chunks of dependent loads, exp and reduction: 400 us for ONE query at 3 030 x 256): (A) all threads, one
// row each: running sum and its exp() -> ev[t]; (B) one wave per class adds the class's ev in the order the single-phase
// kernel used (lane l takes rows l, l + 64, ...; then the wave tree), so the outputs are bit-for-bit what they were;
// (C) one wave does the first-maximum / threshold / count bookkeeping in parallel (first maximum = larger value, then
// lower class).
constexpr int kSeqBlock = 1024;
constexpr int64_t kSeqParRows = 65536;   // up to here the sequential PNN runs as k_cls_pnn_seq_par (below)
__global__ void __launch_bounds__(kSeqBlock) k_cls_pnn_seq(const double* __restrict__ cs, int nq, int nchunks, double* __restrict__ dist,
                                                            double* __restrict__ ev, const int32_t* __restrict__ class_off, int64_t nt,
                                                            int num_classes, int d, double var, double den, int32_t* __restrict__ best_class,
                                                            int32_t* __restrict__ chunks_out, unsigned long long* ticket_word = nullptr,
                                                            unsigned long long ticket = 0) {
    extern __shared__ __attribute__((aligned(16))) double outputs[];
    int* checked = (int*)(outputs + num_classes);
    int* off = checked + num_classes;                                               // class_off, [num_classes + 1]
    __shared__ int best_s, stop_s;
    const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwaves = kSeqBlock / 64;
    double* dq = dist + (size_t)q * nt;
    double* eq = ev + (size_t)q * nt;
    for (int64_t t = threadIdx.x; t < nt; t += kSeqBlock) dq[t] = 0.0;             // distances[i][t] = 0 (:238-241)
    for (int c = threadIdx.x; c < num_classes; c += kSeqBlock) { checked[c] = 1; outputs[c] = 0.0; }
    for (int c = threadIdx.x; c <= num_classes; c += kSeqBlock) off[c] = class_off[c];
    if (threadIdx.x == 0) { best_s = -1; stop_s = 0; }
    __syncthreads();
    // den = total_training_size (:244): the rows held, unless fir_cls_set_total_training_size said otherwise
    int used = 0;
    for (int ch = 0; ch < nchunks; ++ch) {
        ++used;
        int max_fi = (ch + 1) * 32;                                                 // delta_features_count = 32 (:182,247-249)
        if (max_fi > d) max_fi = d;
        const double* csq = cs + ((size_t)ch * nq + q) * nt;
        // (A) rows in parallel
        for (int64_t t = threadIdx.x; t < nt; t += kSeqBlock) {
            int lo = 0, hi = num_classes;                                           // class of row t: off[lo] <= t < off[lo + 1]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (off[mid] <= t) lo = mid; else hi = mid;
            }
            if (checked[lo]) {                                                      // :251 (a dropped class never comes back)
                const double v = dq[t] + csq[t];                                    // distances[i][t] += diff*diff ... (:264)
                dq[t] = v;
                eq[t] = exp(-v / (2 * var * max_fi));                               // :266
            }
        }
        __syncthreads();
        // (B) one wave per class
        for (int c = wave; c < num_classes; c += nwaves) {
            if (!checked[c]) continue;
            double acc = 0.0;
            for (int t = off[c] + lane; t < off[c + 1]; t += 64) acc += eq[t];
            acc = wave_sum(acc);
            if (lane == 0) outputs[c] = acc / den;                                  // :268
        }
        __syncthreads();
        // (C) bookkeeping, one wave
        if (wave == 0) {
            double mx = -DBL_MAX;
            int bi = 0x7FFFFFFF;
            for (int i = lane; i < num_classes; i += 64)
                if (checked[i] && mx < outputs[i]) { mx = outputs[i]; bi = i; }     // :272-279, this lane's first maximum
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const double om = __shfl_xor(mx, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (om > mx || (om == mx && oi < bi)) { mx = om; bi = oi; }
            }
            const int best = bi != 0x7FFFFFFF ? bi : best_s;                        // nothing exceeded -DBL_MAX: the previous best stays
            const float output_threshold = (float)(mx / 1000000000);                // output_dividor = 1E9 (:186,282)
            int variants = 0;
            for (int i = lane; i < num_classes; i += 64)
                if (checked[i]) {
                    if (outputs[i] < output_threshold) checked[i] = 0;              // :285-286
                    else ++variants;
                }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) variants += __shfl_xor(variants, o, 64);
            if (lane == 0) { best_s = best; stop_s = variants == 1; }               // :291
        }
        __syncthreads();
        if (stop_s) break;
    }
    if (threadIdx.x == 0) {
        best_class[q] = best_s;
        chunks_out[q] = used;
        if (ticket_word) {                                                          // one-query call: see k_cls_argbest
            __threadfence_system();
            __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The sequential PNN with its heavy part made parallel. Which classes are still checked only decides which outputs are LOOKED
// AT (:251, :272-286); a checked class's output after chunk ch -- the sum over its rows of exp(-(running sum)/(2 var max_fi))
// -- does not depend on what happened to the other classes. So one wave per (class, chunk, query) computes that output
// outright (lane l: rows l, l + 64, ... of the class, each row's running sum rebuilt from the chunk sums in chunk order;
// then the wave tree -- the summation order of k_cls_pnn_seq), 808 independent waves at 101 classes x 8 chunks instead of
// one workgroup walking the chunks, and one wave then runs the reference's bookkeeping over the chunks (first maximum,
// threshold, drop, count, stop) on the finished table.
// k_cls_pnn_seq_par: grid (num_classes, nchunks, nq) x 64 threads. k_cls_pnn_seq_walk: one wave per query, after it on the
// stream (a device-wide fence + arrival counter inside one kernel cost more than the second launch: 808 L2 write-backs).
// Dynamic LDS of the walk: num_classes ints (+ nchunks x num_classes doubles: table_in_lds).
__global__ void __launch_bounds__(64) k_cls_pnn_seq_par(const double* __restrict__ cs, int nq, int nchunks, const int32_t* __restrict__ class_off,
                                                         int64_t nt, int num_classes, int d, double var, double den, double* __restrict__ table) {
    const int c = blockIdx.x, ch = blockIdx.y, q = blockIdx.z, lane = threadIdx.x;
    int max_fi = (ch + 1) * 32;                                                         // delta_features_count = 32 (:182,247-249)
    if (max_fi > d) max_fi = d;
    double acc = 0.0;
    const int t0 = class_off[c], t1 = class_off[c + 1];
    for (int t = t0 + lane; t < t1; t += 64) {
        double v = 0.0;                                                                 // distances[i][t] = 0 (:238-241)
        for (int k0 = 0; k0 <= ch; k0 += 8) {                                           // eight chunk sums in flight, added in chunk order
            double part[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) part[k] = k0 + k <= ch ? cs[((size_t)(k0 + k) * nq + q) * nt + t] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (k0 + k <= ch) v = v + part[k];                                      // += the chunk's sum (:264)
        }
        acc += exp(-v / (2 * var * max_fi));                                            // :266
    }
    acc = wave_sum(acc);
    if (lane == 0) table[((size_t)q * nchunks + ch) * num_classes + c] = acc / den;     // :268
}

__global__ void __launch_bounds__(64) k_cls_pnn_seq_walk(const double* __restrict__ table, int nchunks, int num_classes,
                                                          int32_t* __restrict__ best_class, int32_t* __restrict__ chunks_out, int table_in_lds,
                                                          unsigned long long* ticket_word, unsigned long long ticket) {
    extern __shared__ __attribute__((aligned(16))) int checked_l[];
    const int q = blockIdx.x, lane = threadIdx.x;
    // the last wave of query q: every output is in `table`; it is copied to LDS in one go when it fits (table_in_lds), so
    // that the walk over the chunks below pays no global round trip per chunk
    for (int i = lane; i < num_classes; i += 64) checked_l[i] = 1;
    int best_s = -1, used = 0;
    const double* tq = table + (size_t)q * nchunks * num_classes;
    if (table_in_lds) {
        double* tl = (double*)(checked_l + ((num_classes + 1) & ~1));
        for (int i = lane; i < nchunks * num_classes; i += 64) tl[i] = tq[i];
        tq = tl;
    }
    for (int chn = 0; chn < nchunks; ++chn) {
        ++used;
        const double* o = tq + (size_t)chn * num_classes;
        double mx = -DBL_MAX;
        int bi = 0x7FFFFFFF;
        for (int i = lane; i < num_classes; i += 64)
            if (checked_l[i] && mx < o[i]) { mx = o[i]; bi = i; }                       // :272-279, this lane's first maximum
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) {
            const double om = __shfl_xor(mx, s, 64);
            const int oi = __shfl_xor(bi, s, 64);
            if (om > mx || (om == mx && oi < bi)) { mx = om; bi = oi; }
        }
        if (bi != 0x7FFFFFFF) best_s = bi;                                              // nothing exceeded -DBL_MAX: the previous best stays
        const float output_threshold = (float)(mx / 1000000000);                        // output_dividor = 1E9 (:186,282)
        int variants = 0;
        for (int i = lane; i < num_classes; i += 64)
            if (checked_l[i]) {
                if (o[i] < output_threshold) checked_l[i] = 0;                          // :285-286
                else ++variants;
            }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) variants += __shfl_xor(variants, s, 64);
        if (variants == 1) break;                                                       // :291
    }
    if (lane == 0) {
        best_class[q] = best_s;
        chunks_out[q] = used;
        if (ticket_word) {
            __threadfence_system();
            __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// mode 0: PNN arg-max, first maximum from -DBL_MAX (classification.cpp:217-224).
// mode 1: kNN arg-min of kth; when no class has k rows the reference's loop ends without a
//         break and the arg-max of the vote counts = the largest class (first) wins (:161-168).
__global__ void __launch_bounds__(64) k_cls_argbest(const double* __restrict__ v, const int32_t* __restrict__ class_off, int num_classes,
                                                     int mode, int32_t* __restrict__ best_class, unsigned long long* ticket_word = nullptr,
                                                     unsigned long long ticket = 0) {
    // ticket_word (one-query calls, results in pinned host memory): after the class, the call's ticket -- the host spins on
    // that word instead of synchronising the stream (cls_wait_ticket)
    // One wave per query: lane l looks at classes l, l + 64, ... in order (its first extremum), then the lanes are folded
    // by (better value, then lower class) -- the first extremum of the reference's single loop.
    const int q = blockIdx.x, lane = threadIdx.x;
    const double* s = v + (size_t)q * num_classes;
    const int kNone = 0x7FFFFFFF;
    int best = kNone;
    if (mode == 0) {
        double mx = -DBL_MAX;
        for (int i = lane; i < num_classes; i += 64)
            if (mx < s[i]) { mx = s[i]; best = i; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double om = __shfl_xor(mx, o, 64);
            const int oi = __shfl_xor(best, o, 64);
            if (oi != kNone && (best == kNone || om > mx || (om == mx && oi < best))) { mx = om; best = oi; }
        }
    } else {
        double mn = DBL_MAX;
        for (int i = lane; i < num_classes; i += 64)
            if (s[i] < mn) { mn = s[i]; best = i; }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const double om = __shfl_xor(mn, o, 64);
            const int oi = __shfl_xor(best, o, 64);
            if (oi != kNone && (best == kNone || om < mn || (om == mn && oi < best))) { mn = om; best = oi; }
        }
        if (best == kNone) {                                  // wave-uniform: no class has k rows
            int mc = -1;
            for (int i = lane; i < num_classes; i += 64) {
                const int cnt = class_off[i + 1] - class_off[i];
                if (cnt > mc) { mc = cnt; best = i; }
            }
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const int om = __shfl_xor(mc, o, 64);
                const int oi = __shfl_xor(best, o, 64);
                if (oi != kNone && (best == kNone || om > mc || (om == mc && oi < best))) { mc = om; best = oi; }
            }
        }
    }
    if (lane != 0) return;
    best_class[q] = best == kNone ? -1 : best;
    if (ticket_word) {
        __threadfence_system();
        __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

struct fir_cls {
    int device = 0, cus = 0;
    int64_t nt = 0;
    int d = 0, dp2 = 0, num_classes = 0;
    int64_t tiles = 0;
    double2* gal2 = nullptr;
    double* avg = nullptr;
    int32_t* class_off = nullptr;
    hipStream_t stream = nullptr;
    double* dq = nullptr; size_t dq_cap = 0;
    double* qn = nullptr; size_t qn_cap = 0;
    double* sums = nullptr; size_t sums_cap = 0;
    double* scores = nullptr; size_t scores_cap = 0;
    int32_t* best = nullptr; size_t best_cap = 0;
    void* pin = nullptr;              // pinned, device-visible staging of small calls: queries in, classes (+ chunk counts) out
    unsigned long long ticket = 0;    // one-query calls so far: the word the host waits for
    double total_training_size = 0;   // 0: nt. PNNwithClustering keeps the full size as denominator (classification.cpp:390,393)
    bool profiling = false;           // fir_cls_profile_enable: HIP event pairs around the launches of the distance scan
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double last_bytes = 0.0;          // algorithmic bytes of the last timed launch
    double last_flops = 0.0;          // ... its dot-product flops when it was a matrix-core pass (kNN batches), else 0
    char last_kernel[64] = "";
    // large kNN batches: the matrix cores nominate, float64 re-ranks, the exact scan takes what is not settled (fir_gemm_f64.h)
    fir_gemm* mm = nullptr;           // created on the first such call
    bool mm_failed = false;           // ... or not (rows too long, no memory): the exact scan answers
    int mm_mode = -1;                 // fir_cls_set_knn_mfma: -1 automatic (>= kKnnMfmaQueries queries against a training set streamed from HBM), 0 never, > 0 from that many queries on
    double* qc = nullptr; size_t qc_cap = 0;          // centred queries, row-major
    int32_t* knn_rows = nullptr; size_t knn_rows_cap = 0;
    double* knn_dist = nullptr; size_t knn_dist_cap = 0;
    int32_t* knn_ok = nullptr; size_t knn_ok_cap = 0;
    int64_t mm_queries = 0, mm_unsettled = 0;         // queries that went through the matrix cores / of them sent on to the exact scan
};

extern "C" void fir_set_last_error_(const char* msg);   // fir_capi.hip: feeds fir_last_error()

namespace {

thread_local char g_cls_err[512];

int cls_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_cls_err, sizeof(g_cls_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_cls_err);
    return code;
}

#define CLS_HIP(expr)                                                                                          \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return cls_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,        \
                                              "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// HIP event pair around one launch of the distance scan (no-ops unless fir_cls_profile_enable is on)
void cls_prof(fir_cls* c, int end, double bytes, const char* kernel) {
    if (!c->profiling) return;
    if (!end && c->ev_used + 2 > c->ev.size())
        for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            c->ev.push_back(e);
        }
    if (c->ev_used + 2 > c->ev.size()) return;
    if (!end) {
        (void)hipEventRecord(c->ev[c->ev_used], c->stream);
    } else {
        (void)hipEventRecord(c->ev[c->ev_used + 1], c->stream);
        c->ev_used += 2;
        c->last_bytes = bytes;
        if (kernel) std::snprintf(c->last_kernel, sizeof c->last_kernel, "%s", kernel);
    }
}

template <typename T>
int cls_grow(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return FIR_OK;
    if (p) CLS_HIP(hipFree(p));
    p = nullptr; cap = 0;
    CLS_HIP(hipMalloc((void**)&p, std::max<size_t>(need, 1024) * sizeof(T)));
    cap = std::max<size_t>(need, 1024);
    return FIR_OK;
}

// distance sums of all qb queries into c->sums (device), in passes of kQB queries
// Small calls (what a per-image predict() loop makes): the queries are read from, and the class ids written to, pinned
// host memory by the kernels themselves -- no copy engine, one synchronisation per call.
constexpr size_t kPinQueryBytes = 256 * 1024;
constexpr int kPinResults = 4096;                 // int32 slots: classes [0, 2048), chunk counts [2048, 4096)
bool cls_small(const fir_cls* c, int32_t qb) { return (size_t)qb * c->d * sizeof(double) <= kPinQueryBytes && qb <= kPinResults / 2; }
int cls_ensure_pin(fir_cls* c) {
    if (c->pin) return FIR_OK;
    CLS_HIP(hipHostMalloc(&c->pin, kPinQueryBytes + kPinResults * sizeof(int32_t) + 64, hipHostMallocDefault));
    std::memset((char*)c->pin + kPinQueryBytes + kPinResults * sizeof(int32_t), 0, 64);
    return FIR_OK;
}
int32_t* cls_pin_results(fir_cls* c) { return (int32_t*)((char*)c->pin + kPinQueryBytes); }
unsigned long long* cls_pin_ticket(fir_cls* c) { return (unsigned long long*)((char*)c->pin + kPinQueryBytes + kPinResults * sizeof(int32_t)); }
// Spin (2 ms at most, then the stream synchronisation) until the call's last kernel has written `ticket` to the pinned word:
// cheaper than hipStreamSynchronize for calls that take tens of microseconds (as fir_capi.hip's wait_ticket).
int cls_wait_ticket(fir_cls* c, unsigned long long ticket) {
    volatile unsigned long long* flag = cls_pin_ticket(c);
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket; ++spins) {
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            CLS_HIP(hipStreamSynchronize(c->stream));
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket) return cls_fail(FIR_ERR_HIP, "the result ticket was not published");
            break;
        }
    }
    return FIR_OK;
}
// -> device-visible pointer to the staged queries
int cls_stage_queries(fir_cls* c, const double* queries, int32_t qb, const double** d_q) {
    int rc;
    if (cls_small(c, qb)) {
        if ((rc = cls_ensure_pin(c))) return rc;
        std::memcpy(c->pin, queries, (size_t)qb * c->d * sizeof(double));
        *d_q = (const double*)c->pin;
        return FIR_OK;
    }
    if ((rc = cls_grow(c->dq, c->dq_cap, (size_t)qb * c->d))) return rc;
    CLS_HIP(hipMemcpyAsync(c->dq, queries, (size_t)qb * c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    *d_q = c->dq;
    return FIR_OK;
}

// Queries per internal batch: the distance table (qb x nt doubles) stays under 1 GiB whatever the caller's batch is.
int32_t cls_batch(const fir_cls* c) {
    const int64_t per_query = std::max<int64_t>(c->nt, 1) * (int64_t)sizeof(double);
    // ... and within gridDim.y of the per-(class, query) kernels (65535)
    return (int32_t)std::max<int64_t>(kQBBig, std::min<int64_t>(65528, ((int64_t)1 << 30) / per_query / kQBBig * kQBBig));
}

int cls_scan(fir_cls* c, const double* queries, int32_t qb) {
    int rc;
    const double* dq = nullptr;
    if ((rc = cls_stage_queries(c, queries, qb, &dq))) return rc;
    if ((rc = cls_grow(c->sums, c->sums_cap, (size_t)qb * std::max<int64_t>(c->nt, 1)))) return rc;
    const int kk = c->dp2 * 2;
    const int waves = (int)std::min<int64_t>(std::max<int64_t>((c->tiles + 3) / 4 * 4, 4), (int64_t)c->cus * 16);
    // a training set streamed from HBM takes 8 queries per pass (the pass stays HBM-bound), a cache-resident one 4
    const bool big = (double)c->tiles * 64.0 * c->dp2 * 16.0 > 256.0 * 1024 * 1024;
    const int qbt = big ? kQBBig : kQB;
    if (qb == 1 && !big) {
        hipLaunchKernelGGL(k_cls_scan_one, dim3(waves / 4), dim3(kBlock), (size_t)kk * sizeof(double), c->stream, c->gal2, dq, c->avg, c->nt,
                           (int)c->tiles, c->dp2, c->d, waves, 0, c->dp2, c->sums, c->dp2, (int64_t)0);
        CLS_HIP(hipGetLastError());
        return FIR_OK;
    }
    const size_t lds_tile = (size_t)(kk + 1) * 8 * sizeof(double);
    if (big && lds_tile <= 64 * 1024) {
        // the LDS-tile scan: every tile of eight queries in ONE launch (blockIdx.y), at most 64 tiles per launch; two tiles per read of
        // the training rows when the call has them and both fit LDS (FIR_CLS_ONE_TILE=1: one, for A/B runs)
        const int ntile = (qb + 7) / 8;
        if ((rc = cls_grow(c->qn, c->qn_cap, (size_t)ntile * kk * 8))) return rc;
        hipLaunchKernelGGL(k_cls_prep_query_tiles, dim3((unsigned)(((size_t)ntile * kk * 8 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, dq, qb, c->d,
                           c->dp2, c->avg, c->qn);
        const bool one_tile = fir_knob_("FIR_CLS_ONE_TILE") != nullptr;
        if (const char* form = fir_knob_("FIR_CLS_FORM")) {
            // experiments (profiles/r04_k3_*.txt): "NT,R,U,BLOCK,WPS" = query tiles per read, rows per lane, double2 in flight per row, threads per
            // workgroup, waves per SIMD of the launch bounds -- one of the instantiations below; the sums are the same bits in every form
            int fnt = 2, fr = 1, fu = 8, fb = 256, fw = 2;
            std::sscanf(form, "%d,%d,%d,%d,%d", &fnt, &fr, &fu, &fb, &fw);
            typedef void (*scan_fn)(const double2*, const double*, int64_t, int, int, int, int, double*);
            scan_fn fn = nullptr;
            const char* nm = "?";
#define FIR_CLS_PICK(NT_, R_, U_, B_, W_) if (fnt == NT_ && fr == R_ && fu == U_ && fb == B_ && fw == W_) { fn = k_cls_scan_lds<U_, NT_, B_, R_, W_>; nm = "fir::k_cls_scan_lds<" #U_ ", " #NT_ ", " #B_ ", " #R_ ", " #W_ ">"; }
            FIR_CLS_PICK(2, 1, 8, 256, 2) FIR_CLS_PICK(2, 2, 4, 256, 2) FIR_CLS_PICK(2, 2, 8, 256, 2) FIR_CLS_PICK(2, 2, 4, 512, 2) FIR_CLS_PICK(2, 1, 4, 512, 4)
            FIR_CLS_PICK(3, 1, 8, 512, 2) FIR_CLS_PICK(3, 2, 4, 512, 2) FIR_CLS_PICK(4, 1, 8, 512, 2) FIR_CLS_PICK(4, 2, 4, 512, 2) FIR_CLS_PICK(4, 1, 4, 512, 2)
            FIR_CLS_PICK(2, 4, 2, 256, 2) FIR_CLS_PICK(2, 2, 2, 256, 2) FIR_CLS_PICK(4, 2, 2, 512, 2)
#undef FIR_CLS_PICK
            if (fn && (size_t)fnt * lds_tile <= 150 * 1024) {
                CLS_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                const int wpc = fw * 4;                                          // waves per CU the launch bounds (and LDS) allow
                const int64_t tsteps = (c->tiles + fr - 1) / fr;
                const int wvf = (int)std::min<int64_t>(std::max<int64_t>((tsteps + fb / 64 - 1) / (fb / 64) * (fb / 64), fb / 64), (int64_t)c->cus * wpc);
                for (int t0 = 0; t0 < ntile; t0 += 64) {
                    const int tn = std::min(64, ntile - t0);
                    cls_prof(c, 0, 0.0, nullptr);
                    hipLaunchKernelGGL(fn, dim3(wvf / (fb / 64), (tn + fnt - 1) / fnt), dim3(fb), (size_t)fnt * lds_tile, c->stream, c->gal2, c->qn + (size_t)t0 * kk * 8, c->nt,
                                       (int)c->tiles, c->dp2, wvf, qb - t0 * 8, c->sums + (size_t)t0 * 8 * c->nt);
                    cls_prof(c, 1, (double)((tn + fnt - 1) / fnt) * ((double)c->tiles * 64.0 * c->dp2 * 16.0) + (double)tn * ((double)kk * 64.0 + 8.0 * 8.0 * (double)c->nt), nm);
                }
                CLS_HIP(hipGetLastError());
                return FIR_OK;
            }
        }
        // Forms by tiles of eight queries per read of the training rows (profiles/r04_k3_forms.txt; 1M x 512, 64 queries, kernel ms per call):
        // one tile, one row per lane 5.7 (HBM-bound, 0.72 of the peak); two tiles 4.16 (round 3); two tiles, two rows per lane 3.90; four
        // tiles, two rows per lane 3.74 -- the f64 vector pipes are then 85-90 % busy at the 1.7 GHz the chip holds under this load
        // (profiles/r04_rocprofv3_pmc_k3.json): that, not the 2.4 GHz issue rate, is the roof. Two rows per lane: every broadcast LDS read of a
        // query value serves two rows (the LDS pipe was two thirds busy at the vector pipes' full rate). A call is cut into groups of 4, 2, 1
        // tiles so that no group computes padding tiles.
        typedef void (*scan_fn)(const double2*, const double*, int64_t, int, int, int, int, double*);
        struct Form { scan_fn fn; int nt, r, block, wpc; const char* name; };
        const Form f4 = {k_cls_scan_lds<2, 4, 512, 2, 2>, 4, 2, 512, 8, "fir::k_cls_scan_lds<2, 4, 512, 2, 2>"};
        const Form f2 = {k_cls_scan_lds<4, 2, 256, 2, 2>, 2, 2, 256, 8, "fir::k_cls_scan_lds<4, 2, 256, 2, 2>"};
        const Form f1 = {k_cls_scan_lds<8, 1>, 1, 1, 256, 16, "fir::k_cls_scan_lds<8, 1>"};
        {
            static bool attr_set[64] = {};                     // (per device: the attribute belongs to the device's copy of the code object)
            const int dv = c->device & 63;
            if (!attr_set[dv]) {
                CLS_HIP(hipFuncSetAttribute((const void*)f4.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                CLS_HIP(hipFuncSetAttribute((const void*)f2.fn, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
                attr_set[dv] = true;
            }
        }
        for (int t0 = 0; t0 < ntile;) {
            const int rem = ntile - t0;
            const Form& f = (rem >= 4 && 4 * lds_tile <= 150 * 1024 && !one_tile) ? f4 : (rem >= 2 && 2 * lds_tile <= 150 * 1024 && !one_tile) ? f2 : f1;
            const int groups = std::min(rem / f.nt, 64 / f.nt);                   // at most 64 tiles per launch (blockIdx.y = group of nt tiles)
            const int tn = groups * f.nt;
            const int wpb = f.block / 64;
            const int64_t tsteps = (c->tiles + f.r - 1) / f.r;
            const int wv = (int)std::min<int64_t>(std::max<int64_t>((tsteps + wpb - 1) / wpb * wpb, wpb), (int64_t)c->cus * f.wpc);
            cls_prof(c, 0, 0.0, nullptr);
            hipLaunchKernelGGL(f.fn, dim3(wv / wpb, groups), dim3(f.block), (size_t)f.nt * lds_tile, c->stream, c->gal2, c->qn + (size_t)t0 * kk * 8, c->nt, (int)c->tiles,
                               c->dp2, wv, std::min(qb - t0 * 8, tn * 8), c->sums + (size_t)t0 * 8 * c->nt);
            // algorithmic bytes of the launch: every read of the training rows serves nt tiles of eight queries (+ the query tiles, + the sums written)
            cls_prof(c, 1, (double)groups * ((double)c->tiles * 64.0 * c->dp2 * 16.0) + (double)tn * ((double)kk * 64.0 + 8.0 * 8.0 * (double)c->nt), f.name);
            t0 += tn;
        }
        CLS_HIP(hipGetLastError());
        return FIR_OK;
    }
    for (int q0 = 0; q0 < qb; q0 += qbt) {
        const int nq = std::min(qbt, qb - q0);
        if (big) {
            hipLaunchKernelGGL(k_cls_prep_queries<kQBBig>, dim3((kk * kQBBig + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
                               dq + (size_t)q0 * c->d, nq, c->d, c->dp2, c->avg, c->qn);
            hipLaunchKernelGGL(k_cls_scan<kQBBig>, dim3(waves / 4), dim3(kBlock), 0, c->stream, c->gal2, c->qn, c->nt, (int)c->tiles, c->dp2,
                               c->d, waves, nq, 0, c->dp2, c->sums + (size_t)q0 * c->nt, c->dp2, (int64_t)0);
        } else {
            hipLaunchKernelGGL(k_cls_prep_queries<kQB>, dim3((kk * kQB + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
                               dq + (size_t)q0 * c->d, nq, c->d, c->dp2, c->avg, c->qn);
            hipLaunchKernelGGL(k_cls_scan<kQB>, dim3(waves / 4), dim3(kBlock), 0, c->stream, c->gal2, c->qn, c->nt, (int)c->tiles, c->dp2,
                               c->d, waves, nq, 0, c->dp2, c->sums + (size_t)q0 * c->nt, c->dp2, (int64_t)0);
        }
    }
    CLS_HIP(hipGetLastError());
    return FIR_OK;
}

}  // namespace

extern "C" {

static int cls_create(const double* train_rows, bool rows_on_device, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                      const double* avg, int32_t device, fir_cls** out);

int fir_cls_create(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                   const double* avg, int32_t device, fir_cls** out) {
    return cls_create(train_rows, false, nt, d, train_class, num_classes, avg, device, out);
}

int fir_cls_create_dev(const double* d_train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                       const double* avg, int32_t device, fir_cls** out) {
    return cls_create(d_train_rows, true, nt, d, train_class, num_classes, avg, device, out);
}

static int cls_create(const double* train_rows, bool rows_on_device, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                      const double* avg, int32_t device, fir_cls** out) {
    if (!out) return cls_fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nt < 0 || d <= 0 || num_classes <= 0 || !avg || (nt > 0 && (!train_rows || !train_class)))
        return cls_fail(FIR_ERR_ARG, "bad arguments (nt=%lld d=%d classes=%d)", (long long)nt, d, num_classes);
    if (nt >= ((int64_t)1 << 31) - 64) return cls_fail(FIR_ERR_ARG, "nt too large");
    std::vector<int32_t> off((size_t)num_classes + 1, 0);
    for (int64_t t = 0; t < nt; ++t) {
        const int32_t cl = train_class[t];
        if (cl < 0 || cl >= num_classes || (t > 0 && cl < train_class[t - 1]))
            return cls_fail(FIR_ERR_ARG, "train_class must be non-decreasing in [0,%d) (row %lld)", num_classes, (long long)t);
        off[(size_t)cl + 1]++;
    }
    for (int i = 0; i < num_classes; ++i) off[(size_t)i + 1] += off[(size_t)i];
    int cnt = 0;
    cnt = fir_device_count();        // the guarded first touch of the runtime (fir_runtime_init_)
    if (cnt <= 0) return cls_fail(FIR_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= cnt) return cls_fail(FIR_ERR_NODEVICE, "device %d out of range (%d visible)", device, cnt);
    { const int rc0 = fir_runtime_init_(device); if (rc0) return rc0; }
    hipDeviceProp_t prop;
    CLS_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return cls_fail(FIR_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    fir_cls* c = new (std::nothrow) fir_cls();
    if (!c) return cls_fail(FIR_ERR_NOMEM, "host allocation failed");
    c->device = device;
    c->cus = prop.multiProcessorCount;
    c->nt = nt;
    c->d = d;
    c->dp2 = (d + 1) / 2;
    c->num_classes = num_classes;
    c->tiles = (nt + kTileRows - 1) / kTileRows;
    int rc = FIR_OK;
    double* stage = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    const size_t g2 = (size_t)std::max<int64_t>(c->tiles, 1) * c->dp2 * 64;
    if (e == hipSuccess) e = hipMalloc((void**)&c->gal2, g2 * sizeof(double2));
    if (e == hipSuccess) e = hipMalloc((void**)&c->avg, (size_t)d * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void**)&c->qn, (size_t)c->dp2 * 2 * kQBBig * sizeof(double));
    if (e == hipSuccess) c->qn_cap = (size_t)c->dp2 * 2 * kQBBig;
    if (e == hipSuccess) e = hipMalloc((void**)&c->class_off, ((size_t)num_classes + 1) * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemcpy(c->avg, avg, (size_t)d * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->class_off, off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    const int64_t slab = std::max<int64_t>(kTileRows, ((int64_t)(256u << 20) / ((int64_t)d * 8)) / kTileRows * kTileRows);
    if (e == hipSuccess && nt > 0 && !rows_on_device) e = hipMalloc((void**)&stage, (size_t)std::min<int64_t>(slab, c->tiles * kTileRows) * d * sizeof(double));
    for (int64_t r0 = 0; e == hipSuccess && r0 < nt; r0 += slab) {
        const int64_t have = std::min<int64_t>(slab, nt - r0);
        if (!rows_on_device) e = hipMemcpyAsync(stage, train_rows + r0 * d, (size_t)have * d * sizeof(double), hipMemcpyHostToDevice, c->stream);
        if (e != hipSuccess) break;
        const int64_t total = ((have + kTileRows - 1) / kTileRows) * c->dp2 * 64;
        hipLaunchKernelGGL(k_cls_retile, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           rows_on_device ? train_rows + r0 * d : stage, have, r0, nt, d, c->dp2, c->avg, c->gal2);
        e = hipStreamSynchronize(c->stream);
    }
    if (stage) (void)hipFree(stage);
    if (e != hipSuccess) rc = cls_fail(e == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP, "training-set upload: %s", hipGetErrorString(e));
    if (rc) { fir_cls_destroy(c); return rc; }
    *out = c;
    return FIR_OK;
}

int fir_cls_set_total_training_size(fir_cls* c, int64_t total) {
    if (!c || total < 0) return cls_fail(FIR_ERR_ARG, "bad argument");
    c->total_training_size = (double)total;
    return FIR_OK;
}

int fir_cls_profile_enable(fir_cls* c, int32_t on) {
    if (!c) return cls_fail(FIR_ERR_ARG, "NULL argument");
    c->profiling = on != 0;
    c->ev_used = 0;
    return FIR_OK;
}

int fir_cls_profile_read(fir_cls* c, float* ms, int32_t cap, int32_t* count, double* bytes_per_launch, char* kernel, int32_t kernel_cap) {
    if (!c) return cls_fail(FIR_ERR_ARG, "NULL argument");
    CLS_HIP(hipSetDevice(c->device));
    const int32_t have = (int32_t)(c->ev_used / 2);
    for (int32_t i = 0; i < have; ++i) {
        CLS_HIP(hipEventSynchronize(c->ev[2 * (size_t)i + 1]));
        float t = 0.f;
        CLS_HIP(hipEventElapsedTime(&t, c->ev[2 * (size_t)i], c->ev[2 * (size_t)i + 1]));
        if (ms && i < cap) ms[i] = t;
    }
    if (count) *count = have;
    if (bytes_per_launch) *bytes_per_launch = c->last_bytes;
    if (kernel && kernel_cap > 0) std::snprintf(kernel, (size_t)kernel_cap, "%s", c->last_kernel);
    c->ev_used = 0;
    return FIR_OK;
}

int fir_cls_destroy(fir_cls* c) {
    if (!c) return FIR_OK;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (hipEvent_t e : c->ev) (void)hipEventDestroy(e);
    (void)hipFree(c->gal2); (void)hipFree(c->avg); (void)hipFree(c->class_off); (void)hipFree(c->dq); (void)hipFree(c->qn);
    (void)hipFree(c->sums); (void)hipFree(c->scores); (void)hipFree(c->best);
    if (c->mm) (void)fir_gemm_destroy(c->mm);
    (void)hipFree(c->qc); (void)hipFree(c->knn_rows); (void)hipFree(c->knn_dist); (void)hipFree(c->knn_ok);
    if (c->pin) (void)hipHostFree(c->pin);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return FIR_OK;
}

int fir_cls_distance_sums(fir_cls* c, const double* queries, int32_t qb, double* sums) {
    if (!c || !sums || (qb > 0 && !queries)) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return cls_fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0 || c->nt == 0) return FIR_OK;
    CLS_HIP(hipSetDevice(c->device));
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    CLS_HIP(hipMemcpyAsync(sums, c->sums, (size_t)qb * c->nt * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CLS_HIP(hipStreamSynchronize(c->stream));
    return FIR_OK;
}

int fir_cls_pnn_predict(fir_cls* c, const double* queries, int32_t qb, double var, double* scores, int32_t* best_class) {
    if (!c || (qb > 0 && !queries)) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return cls_fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0) return FIR_OK;
    if (qb > cls_batch(c)) {
        const int32_t b = cls_batch(c);
        for (int32_t q0 = 0; q0 < qb; q0 += b) {
            const int rc0 = fir_cls_pnn_predict(c, queries + (size_t)q0 * c->d, std::min(b, qb - q0), var,
                                                scores ? scores + (size_t)q0 * c->num_classes : nullptr, best_class ? best_class + q0 : nullptr);
            if (rc0) return rc0;
        }
        return FIR_OK;
    }
    CLS_HIP(hipSetDevice(c->device));
    if (var <= 0) { var = 0.00002; if (c->d > 2000) var /= 10; }                // classification.cpp:190-193
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)qb * c->num_classes))) return rc;
    if ((rc = cls_grow(c->best, c->best_cap, (size_t)qb))) return rc;
    const double denom = (double)(2 * (size_t)c->d) * var;                       // 2*num_of_cont_features*var, :213
    hipLaunchKernelGGL(k_cls_pnn, dim3(c->num_classes, qb), dim3(64), 0, c->stream, c->sums, c->class_off, c->nt, c->num_classes, denom,
                       c->total_training_size > 0 ? c->total_training_size : (double)c->nt, c->scores);
    const bool small = cls_small(c, qb);
    int32_t* dbest = small ? cls_pin_results(c) : c->best;
    const bool one = small && qb == 1 && !scores;       // the reference's predict() per test vector: no stream synchronisation
    const unsigned long long ticket = one ? ++c->ticket : 0;
    hipLaunchKernelGGL(k_cls_argbest, dim3(qb), dim3(64), 0, c->stream, c->scores, c->class_off, c->num_classes, 0, dbest,
                       one ? cls_pin_ticket(c) : (unsigned long long*)nullptr, ticket);
    CLS_HIP(hipGetLastError());
    if (scores) CLS_HIP(hipMemcpyAsync(scores, c->scores, (size_t)qb * c->num_classes * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (best_class && !small) CLS_HIP(hipMemcpyAsync(best_class, c->best, (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (one) { if ((rc = cls_wait_ticket(c, ticket))) return rc; }
    else CLS_HIP(hipStreamSynchronize(c->stream));
    if (best_class && small) std::memcpy(best_class, dbest, (size_t)qb * sizeof(int32_t));
    return FIR_OK;
}

// PNN class scores of qb <= *max_batch queries, left on the device (valid until the handle's next call), queued on the
// handle's stream: the row-sharded PNN (fir_shard.hip) adds the shards' partial sums before the arg-max.
int fir_cls_pnn_scores_dev_(fir_cls* c, const double* queries, int32_t qb, double var, double** d_scores, void** stream, int32_t* max_batch) {
    if (!c || !d_scores || !stream) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (max_batch) *max_batch = cls_batch(c);
    if (qb == 0) return FIR_OK;
    if (qb < 0 || qb > cls_batch(c) || !queries) return cls_fail(FIR_ERR_ARG, "bad batch %d (at most %d at a time)", qb, cls_batch(c));
    CLS_HIP(hipSetDevice(c->device));
    if (var <= 0) { var = 0.00002; if (c->d > 2000) var /= 10; }                // classification.cpp:190-193
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)qb * c->num_classes))) return rc;
    const double denom = (double)(2 * (size_t)c->d) * var;
    hipLaunchKernelGGL(k_cls_pnn, dim3(c->num_classes, qb), dim3(64), 0, c->stream, c->sums, c->class_off, c->nt, c->num_classes, denom,
                       c->total_training_size > 0 ? c->total_training_size : (double)c->nt, c->scores);
    CLS_HIP(hipGetLastError());
    *d_scores = c->scores;
    *stream = c->stream;
    return FIR_OK;
}

// The k smallest mean distances per class of qb <= *max_batch queries, [qb][num_classes][k] ascending (DBL_MAX where the class has
// fewer rows here), left on the device and queued on the handle's stream: the row-sharded kNN vote (fir_shard.hip) merges them.
int fir_cls_knn_nearest_dev_(fir_cls* c, const double* queries, int32_t qb, int32_t k, double** d_lists, void** stream, int32_t* max_batch) {
    if (!c || !d_lists || !stream) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (max_batch) *max_batch = cls_batch(c);
    if (qb == 0) return FIR_OK;
    if (qb < 0 || qb > cls_batch(c) || !queries || k < 1 || k > kKMax) return cls_fail(FIR_ERR_ARG, "bad batch %d / k %d", qb, k);
    CLS_HIP(hipSetDevice(c->device));
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)qb * c->num_classes * (1 + (size_t)k)))) return rc;
    double* lists = c->scores + (size_t)qb * c->num_classes;
    hipLaunchKernelGGL(k_cls_knn_kth, dim3(c->num_classes, qb), dim3(64), 0, c->stream, c->sums, c->class_off, c->nt, c->num_classes, c->d, k, c->scores,
                       lists);
    CLS_HIP(hipGetLastError());
    *d_lists = lists;
    *stream = c->stream;
    return FIR_OK;
}

int fir_cls_pnn_predict_seq(fir_cls* c, const double* queries, int32_t qb, double var, int32_t* best_class, int32_t* chunks_out) {
    if (!c || !best_class || (qb > 0 && !queries)) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return cls_fail(FIR_ERR_ARG, "qb < 0");
    if ((size_t)c->num_classes * 16 + 4 > 60 * 1024) return cls_fail(FIR_ERR_ARG, "num_classes=%d too large for the LDS tables", c->num_classes);
    if (qb == 0) return FIR_OK;
    CLS_HIP(hipSetDevice(c->device));
    if (var <= 0) { var = 0.00002; if (c->d > 2000) var /= 10; }                // classification.cpp:229-233
    const int nchunks = (c->d + 31) / 32;
    const int64_t ntp = std::max<int64_t>(c->nt, 1);
    int rc;
    const double* dq = nullptr;
    if ((rc = cls_stage_queries(c, queries, qb, &dq))) return rc;
    if ((rc = cls_grow(c->sums, c->sums_cap, (size_t)(nchunks + 2) * kQB * ntp))) return rc;   // chunk sums + running sums + their exp()
    if ((rc = cls_grow(c->best, c->best_cap, (size_t)2 * std::max(qb, kQB)))) return rc;
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)kQB * nchunks * c->num_classes))) return rc;   // k_cls_pnn_seq_par's table
    const bool small = cls_small(c, qb);
    int32_t* dbest = small ? cls_pin_results(c) : c->best;
    int32_t* dchunks = small ? cls_pin_results(c) + kPinResults / 2 : c->best + std::max(qb, kQB);
    const int kk = c->dp2 * 2;
    const int waves = (int)std::min<int64_t>(std::max<int64_t>((c->tiles + 3) / 4 * 4, 4), (int64_t)c->cus * 16);
    double* run = c->sums + (size_t)nchunks * kQB * ntp;
    const bool one = small && qb == 1;
    const unsigned long long ticket = one ? ++c->ticket : 0;
    for (int q0 = 0; q0 < qb; q0 += kQB) {
        const int nq = std::min(kQB, qb - q0);
        // all 32-feature chunk sums (16 double2 chunks each) from one pass: chunk ch lands at sums + ch * nq * nt
        if (qb == 1) {
            hipLaunchKernelGGL(k_cls_scan_one, dim3(waves / 4), dim3(kBlock), (size_t)kk * sizeof(double), c->stream, c->gal2, dq, c->avg, c->nt,
                               (int)c->tiles, c->dp2, c->d, waves, 0, c->dp2, c->sums, 16, (int64_t)c->nt);
        } else {
            hipLaunchKernelGGL(k_cls_prep_queries<kQB>, dim3((kk * kQB + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream,
                               dq + (size_t)q0 * c->d, nq, c->d, c->dp2, c->avg, c->qn);
            hipLaunchKernelGGL(k_cls_scan<kQB>, dim3(waves / 4), dim3(kBlock), 0, c->stream, c->gal2, c->qn, c->nt, (int)c->tiles, c->dp2, c->d, waves,
                               nq, 0, c->dp2, c->sums, 16, (int64_t)nq * c->nt);
        }
        const size_t flags_lds = ((size_t)c->num_classes + 1) / 2 * 8, table_lds = (size_t)nchunks * c->num_classes * 8;
        const int table_in_lds = flags_lds + table_lds <= 48 * 1024;
        if (c->nt <= kSeqParRows && flags_lds <= 48 * 1024) {
            hipLaunchKernelGGL(k_cls_pnn_seq_par, dim3(c->num_classes, nchunks, nq), dim3(64), 0, c->stream, c->sums, nq, nchunks, c->class_off, c->nt,
                               c->num_classes, c->d, var, c->total_training_size > 0 ? c->total_training_size : (double)c->nt, c->scores);
            hipLaunchKernelGGL(k_cls_pnn_seq_walk, dim3(nq), dim3(64), flags_lds + (table_in_lds ? table_lds : 0), c->stream, c->scores, nchunks,
                               c->num_classes, dbest + q0, dchunks + q0, table_in_lds, one ? cls_pin_ticket(c) : (unsigned long long*)nullptr, ticket);
        }
        else
            hipLaunchKernelGGL(k_cls_pnn_seq, dim3(nq), dim3(kSeqBlock), (size_t)c->num_classes * 16 + 4, c->stream, c->sums, nq, nchunks, run,
                               run + (size_t)kQB * ntp, c->class_off, c->nt, c->num_classes, c->d, var,
                               c->total_training_size > 0 ? c->total_training_size : (double)c->nt, dbest + q0, dchunks + q0,
                               one ? cls_pin_ticket(c) : (unsigned long long*)nullptr, ticket);
    }
    CLS_HIP(hipGetLastError());
    if (one) {
        if ((rc = cls_wait_ticket(c, ticket))) return rc;
        best_class[0] = dbest[0];
        if (chunks_out) chunks_out[0] = dchunks[0];
        return FIR_OK;
    }
    if (!small) {
        CLS_HIP(hipMemcpyAsync(best_class, dbest, (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
        if (chunks_out) CLS_HIP(hipMemcpyAsync(chunks_out, dchunks, (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    }
    CLS_HIP(hipStreamSynchronize(c->stream));
    if (small) {
        std::memcpy(best_class, dbest, (size_t)qb * sizeof(int32_t));
        if (chunks_out) std::memcpy(chunks_out, dchunks, (size_t)qb * sizeof(int32_t));
    }
    return FIR_OK;
}

}  // extern "C"

namespace {
constexpr int kKnnMfmaQueries = 128;       // automatic mode: kNN batches from this size on take the matrix cores (training sets streamed from HBM)

// qc[i][k] = q[i][k] - avg[k]: the query side of normalize() (classification.cpp:103-105, :135), row-major
__global__ void __launch_bounds__(kBlock) k_cls_center_queries(const double* __restrict__ q, int64_t count, int d, const double* __restrict__ avg, double* __restrict__ qc) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o < count) qc[o] = q[o] - avg[o % d];
}

// The vote over a query's K' nearest rows (ascending, exact float64 distances; fir_gemm_f64.h): classification.cpp:154-160 walks the
// sorted rows and stops at the first class with K votes. best[q] <- that class when the walk ends inside the K' rows, the certificate
// holds and no two of the rows looked at (nor the deciding row and its successor) have EQUAL distances -- the reference's std::sort
// leaves the order of equal distances open, and the exact path has its own rule for them --, else -1: the exact scan answers.
__global__ void __launch_bounds__(kBlock) k_cls_knn_vote(const int32_t* __restrict__ rows, const double* __restrict__ dist, const int32_t* __restrict__ ok, int nq, int kp, int k,
                                                          const int32_t* __restrict__ class_off, int num_classes, int32_t* __restrict__ best) {
    const int q = blockIdx.x * kBlock + threadIdx.x;
    if (q >= nq) return;
    int res = -1;
    if (ok[q]) {
        int cls[8];
        for (int j = 0; j < kp; ++j) {
            const int r = rows[(size_t)q * kp + j];
            if (r < 0) break;
            if (j > 0 && dist[(size_t)q * kp + j] == dist[(size_t)q * kp + j - 1]) break;          // equal distances: not ours to order
            int lo = 0, hi = num_classes;                                                         // class_off[lo] <= r < class_off[hi]
            while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (class_off[mid] <= r) lo = mid; else hi = mid; }
            cls[j] = lo;
            int votes = 0;
            for (int i = 0; i <= j; ++i) votes += cls[i] == lo ? 1 : 0;
            if (votes >= k) {
                // (the successor of the deciding row: inside the K' rows it is known; past them the certificate says nothing ties with the K'-th)
                const bool tie_next = j + 1 < kp && rows[(size_t)q * kp + j + 1] >= 0 && dist[(size_t)q * kp + j + 1] == dist[(size_t)q * kp + j];
                if (!tie_next) res = lo;
                break;
            }
        }
    }
    best[q] = res;
}

// kNN through the matrix cores for the queries of one call (qb <= cls_batch): 0 = best_class filled for every query, 1 = not taken (the
// exact path answers the call), < 0 = error.
int cls_knn_mfma(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class);
}  // namespace

extern "C" {

int fir_cls_set_knn_mfma(fir_cls* c, int32_t min_queries) {
    if (!c) return cls_fail(FIR_ERR_ARG, "NULL argument");
    c->mm_mode = min_queries < 0 ? -1 : min_queries;
    if (min_queries == 0 && c->mm) { (void)fir_gemm_destroy(c->mm); c->mm = nullptr; }
    c->mm_failed = false;
    return FIR_OK;
}

int fir_cls_knn_stats(fir_cls* c, int64_t* matrix_core_queries, int64_t* exact_scan_queries_of_them) {
    if (!c) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (matrix_core_queries) *matrix_core_queries = c->mm_queries;
    if (exact_scan_queries_of_them) *exact_scan_queries_of_them = c->mm_unsettled;
    return FIR_OK;
}

int fir_cls_last_dispatch(fir_cls* c, char* kernel, int32_t kernel_cap, double* bytes_per_launch, double* flops_per_launch) {
    if (!c) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (kernel && kernel_cap > 0) std::snprintf(kernel, (size_t)kernel_cap, "%s", c->last_kernel);
    if (bytes_per_launch) *bytes_per_launch = c->last_bytes;
    if (flops_per_launch) *flops_per_launch = c->last_flops;
    return FIR_OK;
}

static int cls_knn_exact(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class);

int fir_cls_knn_predict(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class) {
    if (!c || !best_class || (qb > 0 && !queries)) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return cls_fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > kKMax) return cls_fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kKMax);
    if (qb == 0) return FIR_OK;
    // Large batches against a training set that streams from HBM: the matrix cores nominate the K' nearest rows, float64 re-ranks them,
    // the vote is taken over those; what that does not settle falls through to the exact scan below (fir_gemm_f64.h). Same classes.
    const bool big = (double)c->tiles * 64.0 * c->dp2 * 16.0 > 256.0 * 1024 * 1024;
    const bool want = c->mm_mode > 0 ? qb >= c->mm_mode : (c->mm_mode < 0 && big && qb >= kKnnMfmaQueries);
    if (want && !c->mm_failed && c->nt > 0) {
        const int32_t b = 16384;                   // queries per internal batch of the matrix-core path (its scratch: two copies of the queries, 8 rows + distances per query)
        bool all = true;
        for (int32_t q0 = 0; q0 < qb && all; q0 += b) {
            const int rc0 = cls_knn_mfma(c, queries + (size_t)q0 * c->d, std::min(b, qb - q0), k, best_class + q0);
            if (rc0 < 0) return rc0;
            if (rc0 == 1) { if (q0 != 0) return cls_fail(FIR_ERR_STATE, "the matrix-core kNN path stopped in the middle of a call"); all = false; }
        }
        if (all) return FIR_OK;
    }
    return cls_knn_exact(c, queries, qb, k, best_class);
}

static int cls_knn_exact(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class) {
    if (qb > cls_batch(c)) {
        const int32_t b = cls_batch(c);
        for (int32_t q0 = 0; q0 < qb; q0 += b) {
            const int rc0 = cls_knn_exact(c, queries + (size_t)q0 * c->d, std::min(b, qb - q0), k, best_class + q0);
            if (rc0) return rc0;
        }
        return FIR_OK;
    }
    CLS_HIP(hipSetDevice(c->device));
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)qb * c->num_classes))) return rc;
    if ((rc = cls_grow(c->best, c->best_cap, (size_t)qb))) return rc;
    hipLaunchKernelGGL(k_cls_knn_kth, dim3(c->num_classes, qb), dim3(64), 0, c->stream, c->sums, c->class_off, c->nt, c->num_classes, c->d, k, c->scores,
                       (double*)nullptr);
    const bool small = cls_small(c, qb);
    int32_t* dbest = small ? cls_pin_results(c) : c->best;
    const bool one = small && qb == 1;
    const unsigned long long ticket = one ? ++c->ticket : 0;
    hipLaunchKernelGGL(k_cls_argbest, dim3(qb), dim3(64), 0, c->stream, c->scores, c->class_off, c->num_classes, 1, dbest,
                       one ? cls_pin_ticket(c) : (unsigned long long*)nullptr, ticket);
    CLS_HIP(hipGetLastError());
    if (!small) CLS_HIP(hipMemcpyAsync(best_class, c->best, (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    if (one) { if ((rc = cls_wait_ticket(c, ticket))) return rc; }
    else CLS_HIP(hipStreamSynchronize(c->stream));
    if (small) std::memcpy(best_class, dbest, (size_t)qb * sizeof(int32_t));
    return FIR_OK;
}

}  // extern "C"

namespace {
int cls_knn_mfma(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class) {
    CLS_HIP(hipSetDevice(c->device));
    if (!c->mm) {
        const int rc = fir_gemm_create_f64_(c->device, c->cus, c->stream, c->gal2, c->nt, c->d, c->dp2, &c->mm);
        if (rc) {
            c->mm = nullptr;
            if (c->mm_mode > 0 || (rc != FIR_ERR_ARG && rc != FIR_ERR_NOMEM)) return rc;     // asked for explicitly, or a real failure
            (void)hipGetLastError();
            c->mm_failed = true;                                                             // automatic: this shape (or this much HBM) stays with the exact scan
            return 1;
        }
    }
    const int kp = k == 1 ? 1 : 8;           // rows nominated per query: the walk of kNN-K ends inside the 8 nearest rows unless the classes are very mixed
    int rc;
    if ((rc = cls_grow(c->dq, c->dq_cap, (size_t)qb * c->d))) return rc;
    if ((rc = cls_grow(c->qc, c->qc_cap, (size_t)qb * c->d))) return rc;
    if ((rc = cls_grow(c->knn_rows, c->knn_rows_cap, (size_t)qb * 8))) return rc;
    if ((rc = cls_grow(c->knn_dist, c->knn_dist_cap, (size_t)qb * 8))) return rc;
    if ((rc = cls_grow(c->knn_ok, c->knn_ok_cap, (size_t)qb))) return rc;
    if ((rc = cls_grow(c->best, c->best_cap, (size_t)qb))) return rc;
    CLS_HIP(hipMemcpyAsync(c->dq, queries, (size_t)qb * c->d * sizeof(double), hipMemcpyHostToDevice, c->stream));
    const int64_t count = (int64_t)qb * c->d;
    hipLaunchKernelGGL(k_cls_center_queries, dim3((unsigned)((count + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, c->dq, count, c->d, c->avg, c->qc);
    hipEvent_t* evp = nullptr;
    if (c->profiling) {
        if (c->ev_used + 2 > c->ev.size())
            for (int i = 0; i < 64; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) break;
                c->ev.push_back(e);
            }
        if (c->ev_used + 2 <= c->ev.size()) evp = &c->ev[c->ev_used];
    }
    const char* kname = nullptr;
    double flops = 0.0;
    if ((rc = fir_gemm_knn_f64_(c->mm, c->qc, qb, kp, c->knn_rows, c->knn_dist, c->knn_ok, c->stream, &kname, &flops, evp))) return rc;
    if (evp) {
        c->ev_used += 2;
        c->last_bytes = 0.0;
        c->last_flops = flops;
        if (kname) std::snprintf(c->last_kernel, sizeof c->last_kernel, "%s", kname);
    }
    hipLaunchKernelGGL(k_cls_knn_vote, dim3((qb + kBlock - 1) / kBlock), dim3(kBlock), 0, c->stream, c->knn_rows, c->knn_dist, c->knn_ok, qb, kp, k, c->class_off, c->num_classes,
                       c->best);
    CLS_HIP(hipGetLastError());
    CLS_HIP(hipMemcpyAsync(best_class, c->best, (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    CLS_HIP(hipStreamSynchronize(c->stream));
    c->mm_queries += qb;
    // what the K' rows did not settle: the exact scan (classification.cpp's own loop over every row), a tile of queries at a time
    std::vector<int32_t> which;
    for (int32_t i = 0; i < qb; ++i)
        if (best_class[i] < 0) which.push_back(i);
    if (!which.empty()) {
        c->mm_unsettled += (int64_t)which.size();
        std::vector<double> sub(which.size() * (size_t)c->d);
        for (size_t i = 0; i < which.size(); ++i) std::memcpy(&sub[i * (size_t)c->d], queries + (size_t)which[i] * c->d, (size_t)c->d * sizeof(double));
        std::vector<int32_t> cls(which.size());
        char keep[sizeof c->last_kernel];
        std::memcpy(keep, c->last_kernel, sizeof keep);
        const double keep_flops = c->last_flops;
        if ((rc = cls_knn_exact(c, sub.data(), (int32_t)which.size(), k, cls.data()))) return rc;
        for (size_t i = 0; i < which.size(); ++i) best_class[which[i]] = cls[i];
        if (evp) {                                       // the call's dominant kernel stays the matrix-core pass
            std::memcpy(c->last_kernel, keep, sizeof keep);
            c->last_flops = keep_flops;
            c->last_bytes = 0.0;
        }
    }
    return 0;
}
}  // namespace

extern "C" {

int fir_cls_knn_class_nearest(fir_cls* c, const double* queries, int32_t qb, int32_t k, double* nearest) {
    if (!c || !nearest || (qb > 0 && !queries)) return cls_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return cls_fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > kKMax) return cls_fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kKMax);
    if (qb == 0) return FIR_OK;
    const size_t per_query = (size_t)c->num_classes * k;
    if (qb > cls_batch(c)) {
        const int32_t b = cls_batch(c);
        for (int32_t q0 = 0; q0 < qb; q0 += b) {
            const int rc0 = fir_cls_knn_class_nearest(c, queries + (size_t)q0 * c->d, std::min(b, qb - q0), k, nearest + (size_t)q0 * per_query);
            if (rc0) return rc0;
        }
        return FIR_OK;
    }
    CLS_HIP(hipSetDevice(c->device));
    int rc = cls_scan(c, queries, qb);
    if (rc) return rc;
    // scores: [qb][num_classes] k-th values, then [qb][num_classes][k] lists
    if ((rc = cls_grow(c->scores, c->scores_cap, (size_t)qb * c->num_classes * (1 + (size_t)k)))) return rc;
    double* lists = c->scores + (size_t)qb * c->num_classes;
    hipLaunchKernelGGL(k_cls_knn_kth, dim3(c->num_classes, qb), dim3(64), 0, c->stream, c->sums, c->class_off, c->nt, c->num_classes, c->d, k, c->scores,
                       lists);
    CLS_HIP(hipGetLastError());
    CLS_HIP(hipMemcpyAsync(nearest, lists, (size_t)qb * per_query * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CLS_HIP(hipStreamSynchronize(c->stream));
    return FIR_OK;
}

}  // extern "C"
