// fir_kernels.h -- gfx950 (MI355X / CDNA4) device code of the gallery matcher.
//
// One idea carries every kernel here: the gallery is re-tiled ONCE, at create time, into
//
//     tile t (64 consecutive rows)  x  chunk c (4 consecutive features)  x  lane r (row in tile)
//     float4 at  gal4[(t * dp4 + c) * 64 + r]            dp4 = ceil(d / 4)
//
// so that a wavefront owns 64 rows, lane r owns row 64 t + r, and the wave's
// `global_load_dwordx4` for chunk c reads one contiguous, perfectly coalesced KiB.
// A tile is one linear 64*dp4*16-byte stream (128 KiB at d = 512) that goes straight into
// VGPRs: no LDS staging, no transposition, no cross-lane reduction per row. Each lane then
// accumulates ITS row's distance to QB queries feature by feature, in ascending feature order,
// with one IEEE rounding per operation -- exactly the evaluation order of the reference's scalar
// loop (qt_cpp/db_features.cpp:22-42), which makes L2 and chi-square results bit-identical to
// the reference, not merely close. The query values are wave-uniform and are fetched through the
// scalar cache (s_load from a [feature][QB] transposed tile) as SGPR operands of the VALU ops.
//
// Roofline: the scan reads n*d*4 gallery bytes once per pass for QB queries; it is HBM-bound
// while 3*QB VALU ops per gallery float (L2, un-fused sub/mul/add) fit under the HBM time
// (QB <= 8 at d = 512 on MI355X); chi-square / KL are VALU-bound (IEEE division / log per
// element) and are priced against the vector-ALU roof instead.
#pragma once
#include <type_traits>

#include "fir_common.h"

namespace fir {

#ifndef FIR_PIPE
#define FIR_PIPE 0
#endif
#ifndef FIR_NT
#define FIR_NT 1   // +6 % on the 1M x 512 scan (profiles/r01_sweep_notes.md)
#endif
// One lane's float4 of a gallery tile. nt: non-temporal hint -- a gallery that is streamed from HBM once per pass
// should not displace anything in L2; a gallery small enough to live in the L2s keeps the plain load so that the
// next call finds it there (ScanArgs::nt, set per gallery by its size).
__device__ __forceinline__ float4 ld_gallery(const float4* p, bool nt) {
#if FIR_NT
    if (nt) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f v = __builtin_nontemporal_load((const v4f*)p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
#endif
    return *p;
}
// U chunks of one lane, `stride` float4 apart; one uniform branch for the whole group.
template <int U>
__device__ __forceinline__ void ld_gallery_group(float4 (&g)[U], const float4* p, bool nt) {
#if FIR_NT
    if (nt) {
        typedef float v4f __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const v4f v = __builtin_nontemporal_load((const v4f*)(p + (size_t)u * 64));
            g[u] = make_float4(v.x, v.y, v.z, v.w);
        }
        return;
    }
#endif
#pragma unroll
    for (int u = 0; u < U; ++u) g[u] = p[(size_t)u * 64];
}

struct ScanArgs {
    const float4* gal4;   // tiled gallery
    const float* qt;      // query tile, transposed: qt[k * QB + q], k in [0, dp4*4)
    int64_t n;            // rows in this gallery (shard)
    int32_t tiles;        // ceil(n / 64)
    int32_t dp4;          // float4 chunks per row
    int32_t start, end;   // feature range [start, end)
    int32_t waves;        // total waves in the grid
    int64_t row_offset;   // global index of local row 0
    uint64_t* keys;       // kEpiTop1: [QB] packed keys (pre-set to kKeyNone); kEpiTopK: partials [waves][QB][K]
    float* out;           // kEpiStore: out[q * out_stride + row]
    int64_t out_stride;
    int32_t nq;           // live queries in this tile (<= QB); only kEpiStore needs it
    int32_t k;            // kEpiTopK
    int64_t qt_stride;    // floats between the query tiles of consecutive blockIdx.y (top-1 kernels)
    int32_t nt;           // non-temporal gallery loads (galleries that do not fit the L2s)
    int32_t step;         // k_scan_subranges: features per sub-range
    int32_t step2;        // ... != 0: the first sub-range has `step` features, every later one `step2` (the two stages of the conventional TWD)
    const float* tau;     // append form of the L2 scan: rows with distance <= tau[query] are appended ...
    int32_t* counts;      // ... counts[query] entries so far, lists = keys[query * k + slot] (k = capacity per query)
    uint64_t* publish;    // top-1, whole call = ONE launch: the last workgroup to finish writes keys[0..QB) to publish[0..QB) (pinned
    int32_t* done;        // host memory), then `ticket` to publish[QB] -- the host spins on that word instead of synchronising
    uint64_t ticket;      // the stream -- and re-arms keys and the done counter for the next call
    const int32_t* range; // chi-square / KL: range[0] != 0 = some gallery value is outside in_plain_range(), range[1] = serial
    int32_t serial;       // of the last query transposition that met one; NULL or a match = IEEE division sequence
    int32_t* flag;        // kChi2Approx (nomination scan): raised when the operands are not all in the plain range -- the caller falls back
    const float* sg;      // kChi2Harm: sg[row] = sum of the row's values over [start, end) ...
    const float* sq;      // ... sq[query] = the same for the queries of the call (indexed like tau)
    int32_t groups;       // k_scan, top-1 epilogue, > 1: the tiles form `groups` disjoint sets (tile index mod groups; a.waves is a multiple of it, so
    int64_t group_stride; // a wave stays inside one set) and set s reports into keys + s * group_stride: the K row samples of topk_lists_dev in ONE launch
};

template <int QB, int METRIC, int U>
struct TileAcc {
    static constexpr int kSq = 4 * QB;   // query values one chunk needs: 4 features x QB queries

    template <typename QP>
    static __device__ __forceinline__ void load_sq(float (&sq)[kSq], QP qc, int c) {
#pragma unroll
        for (int i = 0; i < kSq; ++i) sq[i] = qc[c * kSq + i];   // uniform address, constant AS: s_load_dwordx8/x16
    }
    static __device__ __forceinline__ void chunk(float (&acc)[QB], const float4 g, const float (&sq)[kSq]) {
        const float gv[4] = {g.x, g.y, g.z, g.w};
        if constexpr (METRIC == kChi2Harm && (QB % 2) == 0) {
            // chi-square = sum(l) + sum(r) - 4 sum_k 1/(1/l_k + 1/r_k). With A = 1/l_j + 1/r_j and B the same for feature j + 1, the two
            // harmonic terms share ONE reciprocal: 1/A + 1/B = (A + B) / (A B) -- per pair of features and pair of queries four packed
            // adds / multiplies, two v_rcp_f32 and one packed fma = 2.25 issue slots per (value, query) (one reciprocal per term: 3;
            // kChi2Approx: 5.5-6.6), plus 1/r once per gallery value and pass. sq holds 1/l, 2^60 for l = 0 and for the padding, and 1/r
            // is held to 2^60 as well: the term is then <= 2^-60 where the reference skips l + r = 0 (inf would make A B rcp(A B) a NaN);
            // plain-range operands keep A B within [2^-30, 2^122].
            typedef float f2v __attribute__((ext_vector_type(2)));
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = __builtin_fminf(__builtin_amdgcn_rcpf(gv[j]), 0x1p60f);
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const f2v va = {v[j], v[j]}, vb = {v[j + 1], v[j + 1]};
#pragma unroll
                for (int p = 0; p < QB / 2; ++p) {
                    const f2v ua = {sq[j * QB + 2 * p], sq[j * QB + 2 * p + 1]};
                    const f2v ub = {sq[(j + 1) * QB + 2 * p], sq[(j + 1) * QB + 2 * p + 1]};
                    const f2v A = ua + va, B = ub + vb;
                    const f2v S = A + B, P = A * B;
                    const f2v R = {__builtin_amdgcn_rcpf(P.x), __builtin_amdgcn_rcpf(P.y)};
                    f2v a2 = {acc[2 * p], acc[2 * p + 1]};
                    a2 = __builtin_elementwise_fma(S, R, a2);
                    acc[2 * p] = a2.x;
                    acc[2 * p + 1] = a2.y;
                }
                if constexpr (QB >= 4) __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if constexpr (METRIC == kKLEnt && (QB % 2) == 0) {
            // sum_k (l_k + r_k) log2 (l_k + r_k) two queries at a time: v_pk_add, two v_log_f32, v_pk_fma = 3 issue slots per
            // (value, query) (the exact form: two quotients and two logarithms, ~35)
            typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2v g2 = {gv[j], gv[j]};
#pragma unroll
                for (int p = 0; p < QB / 2; ++p) {
                    const f2v l2 = {sq[j * QB + 2 * p], sq[j * QB + 2 * p + 1]};
                    const f2v s2 = l2 + g2;
                    const f2v lg = {__builtin_amdgcn_logf(s2.x), __builtin_amdgcn_logf(s2.y)};
                    f2v a2 = {acc[2 * p], acc[2 * p + 1]};
                    a2 = __builtin_elementwise_fma(s2, lg, a2);
                    acc[2 * p] = a2.x;
                    acc[2 * p + 1] = a2.y;
                }
                if constexpr (QB >= 4) __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if constexpr (METRIC == kChi2Approx && (QB % 2) == 0) {
            // the nomination metric two queries at a time: v_pk_add (l - r, by neg), v_pk_add (l + r), 2 v_max, 2 v_rcp, v_pk_mul x 2,
            // v_pk_add = 5.5 issue slots per element (the scalar form the compiler finds: 7.2; the exact division sequence: 11)
            typedef float f2v __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f2v g2 = {gv[j], gv[j]};
#pragma unroll
                for (int p = 0; p < QB / 2; ++p) {
                    const f2v l2 = {sq[j * QB + 2 * p], sq[j * QB + 2 * p + 1]};
                    const f2v df = l2 - g2;
                    const f2v s2 = l2 + g2;
                    const f2v r2 = {__builtin_amdgcn_rcpf(__builtin_fmaxf(s2.x, 0x1p-60f)), __builtin_amdgcn_rcpf(__builtin_fmaxf(s2.y, 0x1p-60f))};
                    const f2v t2 = (df * df) * r2;
                    f2v a2 = {acc[2 * p], acc[2 * p + 1]};
                    a2 = a2 + t2;
                    acc[2 * p] = a2.x;
                    acc[2 * p + 1] = a2.y;
                }
                if constexpr (QB >= 4) __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] = accum<METRIC>(acc[q], sq[j * QB + q], gv[j]);
            // plain-range forms: QB independent division chains (QB/2 packed) are enough to hide the pipeline latency;
            // left alone the scheduler interleaves all 4*QB chains of the chunk and spills
            if constexpr (METRIC >= kChi2InRange && QB >= 4) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // acc[q] += contribution of one group of U chunks starting at chunk c. QP = constant-address-space pointer: the
    // query values come through the scalar cache (s_load); the schedule barrier after each chunk keeps the compiler
    // from hoisting a whole group's scalar loads at once (4*QB SGPRs per chunk; hoisting them all spills SGPRs).
    // QP = plain pointer into LDS: broadcast ds_reads, in-order and counted, scheduled by the compiler.
    template <typename QP>
    static __device__ __forceinline__ void group(float (&acc)[QB], const float4 (&g)[U], QP qc, int c) {
        if constexpr (METRIC == kKLEnt) {
            // the entropy terms are large next to the distance they cancel to: a group's 4 U terms are added up on their own and
            // join the running sum once -- 4 U + nf / (4 U) roundings of the sum's magnitude on a value's way instead of nf
            float part[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) part[q] = 0.0f;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float cur[kSq];
                load_sq(cur, qc, c + u);
                chunk(part, g[u], cur);
                if constexpr (sizeof(QP) == sizeof(sfloat_p) && __is_same(QP, sfloat_p)) __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] += part[q];
            return;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float cur[kSq];
            load_sq(cur, qc, c + u);
            chunk(acc, g[u], cur);
            if constexpr (sizeof(QP) == sizeof(sfloat_p) && __is_same(QP, sfloat_p)) __builtin_amdgcn_sched_barrier(0);
        }
    }
    // One chunk with a feature mask [k0, k1) (range edges that are not multiples of 4).
    template <typename QP>
    static __device__ __forceinline__ void masked(float (&acc)[QB], const float4 g, QP qc, int c, int k0, int k1) {
        const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = c * 4 + j;
            if (k >= k0 && k < k1) {   // wave-uniform
#pragma unroll
                for (int q = 0; q < QB; ++q) acc[q] = accum<METRIC>(acc[q], qc[k * QB + q], gv[j]);
            }
        }
    }
};

// Streams whole tiles; lane r of a wave owns row 64 t + r. EPI selects what happens to the
// finished distances: running first-minimum (top-1), running K smallest (top-K), or store.
// LDSQ = 1: the query tile is staged in (dynamic) LDS first -- dp4*4*QB floats -- and read from there; used for
// one- and two-query tiles over galleries of few tiles, where a wave streams a whole tile alone and the per-chunk
// scalar-load latency of the default form is what it waits for.
template <int QB, int METRIC, int U, int EPI, int KMAX, int WPS, int LDSQ = 0>
__global__ void __launch_bounds__(kBlock, WPS) k_scan(const ScanArgs a) {
    if constexpr (METRIC != kL2) {
        // chi-square / KL come as a pair of launches: the kernel whose arithmetic matches the operands of this call runs
        // (every value in the plain range -> the kChi2InRange / kKLInRange form, fir_common.h), the other one returns here
        const bool plain = a.range != nullptr && a.range[0] == 0 && a.range[1] != a.serial;
        if constexpr (METRIC == kChi2Approx || METRIC == kChi2Harm || METRIC == kKLEnt) {
            // launched alone; its error bound needs non-negative, normal operands: otherwise the caller's exact path answers
            if (!plain) {
                if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.flag) atomicOr(a.flag, 1);
                return;
            }
        } else if (plain != (METRIC == kChi2InRange || METRIC == kKLInRange)) return;
    }
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    // blockIdx.y = query tile (top-1 and append forms): several tiles of QB queries share one launch
    const float* qsrc = a.qt + (EPI == kEpiTop1 || EPI == kEpiAppend ? (size_t)blockIdx.y * a.qt_stride : 0);
    extern __shared__ __attribute__((aligned(16))) float lds_q[];
    if constexpr (LDSQ) {
        const int nf = a.dp4 * 4 * QB;
        for (int i = threadIdx.x; i < nf; i += kBlock) lds_q[i] = qsrc[i];
        __syncthreads();
    }
    typedef typename std::conditional<LDSQ != 0, const float*, sfloat_p>::type QP;
    QP qc;
    if constexpr (LDSQ) qc = lds_q;
    else qc = (sfloat_p)(uintptr_t)qsrc;

    const int c_lo = (a.start + 3) >> 2;          // first whole chunk
    const int c_hi = a.end >> 2;                  // one past the last whole chunk
    const float fcount = (float)(a.end - a.start);  // db_features.cpp:40 divides by (end_pos-start_pos)

    float best_d[EPI == kEpiTop1 || EPI == kEpiAppend ? QB : 1];      // kEpiAppend: the thresholds tau
    int32_t best_i[EPI == kEpiTop1 ? QB : 1];
    float kd[EPI == kEpiTopK ? QB : 1][EPI == kEpiTopK ? KMAX : 1];
    int32_t ki[EPI == kEpiTopK ? QB : 1][EPI == kEpiTopK ? KMAX : 1];
    if constexpr (EPI == kEpiTop1) {
#pragma unroll
        for (int q = 0; q < QB; ++q) { best_d[q] = kNotFound; best_i[q] = -1; }
    }
    if constexpr (EPI == kEpiAppend) {
#pragma unroll
        for (int q = 0; q < QB; ++q) best_d[q] = a.tau[(size_t)blockIdx.y * QB + q];
    }
    if constexpr (EPI == kEpiTopK) {
#pragma unroll
        for (int q = 0; q < QB; ++q)
#pragma unroll
            for (int i = 0; i < KMAX; ++i) { kd[q][i] = kNotFound; ki[q][i] = -1; }
    }

    const int ng = c_lo <= c_hi ? (c_hi - c_lo) / U : 0;   // whole groups of U chunks
    const int c_end = c_lo + ng * U;

    for (int t = gw; t < a.tiles; t += a.waves) {
        const float4* tile = a.gal4 + (size_t)t * a.dp4 * 64 + lane;
        float acc[QB];
#pragma unroll
        for (int q = 0; q < QB; ++q) acc[q] = 0.0f;

        if (c_lo > c_hi) {
            // the whole range lies inside one chunk
            TileAcc<QB, METRIC, U>::masked(acc, tile[(size_t)(a.start >> 2) * 64], qc, a.start >> 2, a.start, a.end);
        } else {
            if ((a.start & 3) != 0)
                TileAcc<QB, METRIC, U>::masked(acc, tile[(size_t)(c_lo - 1) * 64], qc, c_lo - 1, a.start, a.end);

            const float4* p = tile + (size_t)c_lo * 64;
#if FIR_PIPE == 0
            for (int gi = 0; gi < ng; ++gi) {
                float4 g[U];
                ld_gallery_group<U>(g, p + (size_t)(gi * U) * 64, a.nt != 0);
                TileAcc<QB, METRIC, U>::group(acc, g, qc, c_lo + gi * U);
            }
#else
            // register double buffer: group gi+1 is in flight while group gi is consumed
            float4 g[U];
            if (ng > 0) {
                ld_gallery_group<U>(g, p, a.nt != 0);
            }
            for (int gi = 0; gi < ng; ++gi) {
                float4 nx[U];
                const int gn = gi + 1 < ng ? gi + 1 : gi;   // the last group re-reads itself (cache hit, keeps the loop branch-free)
                ld_gallery_group<U>(nx, p + (size_t)(gn * U) * 64, a.nt != 0);
                TileAcc<QB, METRIC, U>::group(acc, g, qc, c_lo + gi * U);
#pragma unroll
                for (int u = 0; u < U; ++u) g[u] = nx[u];
            }
#endif
            for (int c = c_end; c < c_hi; ++c)
                TileAcc<QB, METRIC, U>::masked(acc, tile[(size_t)c * 64], qc, c, a.start, a.end);
            if ((a.end & 3) != 0)
                TileAcc<QB, METRIC, U>::masked(acc, tile[(size_t)c_hi * 64], qc, c_hi, a.start, a.end);
        }

        const int64_t row = (int64_t)t * kTileRows + lane;
        if (row < a.n) {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
                float dist = acc[q] / fcount;                            // db_features.cpp:40
                if constexpr (METRIC == kChi2Harm) dist = ((a.sq[(size_t)blockIdx.y * QB + q] + a.sg[row]) - 4.0f * acc[q]) / fcount;
                if constexpr (METRIC == kKLEnt) dist = (0.693147181f * ((a.sq[(size_t)blockIdx.y * QB + q] + a.sg[row]) - acc[q])) / fcount;
                if constexpr (EPI == kEpiTop1) {
                    if (dist < best_d[q]) { best_d[q] = dist; best_i[q] = (int32_t)row; }   // db_features.cpp:329-332
                } else if constexpr (EPI == kEpiTopK) {
                    if (dist < kd[q][KMAX - 1]) {
                        // insert keeping ascending order; strict '<' keeps earlier rows first on ties
                        float cd = dist; int32_t ci = (int32_t)row;
#pragma unroll
                        for (int i = 0; i < KMAX; ++i) {
                            const bool sw = cd < kd[q][i];
                            const float td = kd[q][i]; const int32_t ti = ki[q][i];
                            kd[q][i] = sw ? cd : td; ki[q][i] = sw ? ci : ti;
                            cd = sw ? td : cd; ci = sw ? ti : ci;
                        }
                    }
                } else if constexpr (EPI == kEpiAppend) {
                    // candidate-list form of the top-K scan (see k_scan_l2_lds<..., APPEND>): rows at or below the threshold
                    if (dist <= best_d[q]) {
                        const size_t qy = (size_t)blockIdx.y * QB + q;
                        const int slot = atomicAdd(&a.counts[qy], 1);
                        if (slot < a.k) a.keys[qy * a.k + slot] = key_pack(dist, (uint32_t)(row + a.row_offset));
                    }
                } else {
                    if (q < a.nq) a.out[(size_t)q * a.out_stride + row] = dist;
                }
            }
        }
    }

    if constexpr (EPI == kEpiTop1) {
        uint64_t* keys = a.keys + (size_t)blockIdx.y * QB + (a.groups > 1 ? (size_t)(gw % a.groups) * a.group_stride : 0);
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            uint64_t key = best_i[q] >= 0 ? key_pack(best_d[q], (uint32_t)((int64_t)best_i[q] + a.row_offset)) : kKeyNone;
            key = wave_min_u64(key);
            if (lane == 0 && key != kKeyNone) {
                // most waves lose against what is already there: read first, contend only to win
                const uint64_t cur = __hip_atomic_load(keys + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (key < cur) atomicMin((unsigned long long*)(keys + q), (unsigned long long)key);
            }
        }
        if (a.publish) {
            __shared__ int last_block;
            __threadfence();                                   // this workgroup's minima are out before it is counted
            __syncthreads();
            if (threadIdx.x == 0) last_block = atomicAdd(a.done, 1) == (int)(gridDim.x * gridDim.y) - 1;
            __syncthreads();
            if (last_block && threadIdx.x == 0) {
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    __hip_atomic_store(a.publish + q, atomicExch((unsigned long long*)(keys + q), (unsigned long long)kKeyNone), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_SYSTEM);
                *a.done = 0;
                __hip_atomic_store(a.publish + QB, a.ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);   // after the keys
            }
        }
    }
    if constexpr (EPI == kEpiTopK) {
        // K rounds: the wave's next smallest key strictly greater than the previous winner.
        // Keys are unique (the row index is part of the key), so "greater than" removes exactly one.
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            uint64_t lk[KMAX];
#pragma unroll
            for (int i = 0; i < KMAX; ++i)
                lk[i] = ki[q][i] >= 0 ? key_pack(kd[q][i], (uint32_t)((int64_t)ki[q][i] + a.row_offset)) : kKeyNone;
            uint64_t prev = 0;
            bool first = true;
            for (int r = 0; r < a.k; ++r) {
                uint64_t cand = kKeyNone;
#pragma unroll
                for (int i = KMAX - 1; i >= 0; --i)
                    if (first || lk[i] > prev) cand = lk[i] < cand ? lk[i] : cand;
                const uint64_t w = wave_min_u64(cand);
                if (lane == 0) a.keys[((size_t)gw * QB + q) * a.k + r] = w;
                prev = w;
                first = false;
                if (w == kKeyNone) {
                    for (int r2 = r + 1; r2 < a.k; ++r2)
                        if (lane == 0) a.keys[((size_t)gw * QB + q) * a.k + r2] = kKeyNone;
                    break;
                }
            }
        }
    }
}

// All distances of CONSECUTIVE feature sub-ranges in one gallery pass: sub-range ci = [start + ci*step, start + (ci+1)*step),
// each a fresh sum divided by `step` (the reference's distance(row, cur, cur + delta) of ProposedTWDClassifier,
// ImageTesting.cpp:243-250, and of the second TWD stage) -- out[(ci * k + q) * out_stride + row], k = queries of the whole
// call (this launch handles nq of them; `out` already points at its first one). start, end and step are multiples of
// 4*U features. One read of [start, end) per tile instead of one launch per sub-range.
template <int QB, int METRIC, int U, int WPS>
__global__ void __launch_bounds__(kBlock, WPS) k_scan_subranges(const ScanArgs a) {
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    sfloat_p qc = (sfloat_p)(uintptr_t)a.qt;
    for (int t = gw; t < a.tiles; t += a.waves) {
        const float4* tile = a.gal4 + (size_t)t * a.dp4 * 64 + lane;
        const int64_t row = (int64_t)t * kTileRows + lane;
        int ci = 0, cstep = a.step >> 2;
        for (int c0 = a.start >> 2; c0 < (a.end >> 2); c0 += cstep, ++ci) {
            const int feats = ci > 0 && a.step2 ? a.step2 : a.step;
            cstep = feats >> 2;
            const float fcount = (float)feats;                        // db_features.cpp:40
            float acc[QB];
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[q] = 0.0f;
            for (int c = c0; c < c0 + cstep; c += U) {
                float4 g[U];
                ld_gallery_group<U>(g, tile + (size_t)c * 64, a.nt != 0);
                TileAcc<QB, METRIC, U>::group(acc, g, qc, c);
            }
            if (row < a.n) {
#pragma unroll
                for (int q = 0; q < QB; ++q)
                    if (q < a.nq) a.out[((size_t)ci * a.k + q) * a.out_stride + row] = acc[q] / fcount;
            }
        }
    }
}

// The nomination scans of topk_lists_dev (kChi2Harm, kKLEnt; append epilogue only) with NH tiles of 8 queries per gallery read:
// at 2.25-3 issue slots per (value, query) an 8-query pass over 1M x 512 takes 0.44 ms = 4.7 TB/s of gallery -- the memory system,
// not the vector pipes, is then what a pass waits for; every load group is therefore used for NH query tiles in turn (their values
// still come through the scalar cache, 32 SGPRs per chunk at a time; blockIdx.y = group of NH consecutive tiles).
template <int METRIC, int NH, int U, int WPS>
__global__ void __launch_bounds__(kBlock, WPS) k_nominate(const ScanArgs a) {
    constexpr int QB = 8;
    const bool plain = a.range != nullptr && a.range[0] == 0 && a.range[1] != a.serial;
    if (!plain) {       // the error bounds need non-negative, normal operands: otherwise the caller's exact path answers
        if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && a.flag) atomicOr(a.flag, 1);
        return;
    }
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    sfloat_p qc[NH];
    float tau[NH][QB];
#pragma unroll
    for (int h = 0; h < NH; ++h) {
        qc[h] = (sfloat_p)(uintptr_t)(a.qt + ((size_t)blockIdx.y * NH + h) * a.qt_stride);
#pragma unroll
        for (int q = 0; q < QB; ++q) tau[h][q] = a.tau[((size_t)blockIdx.y * NH + h) * QB + q];
    }
    const int c_lo = (a.start + 3) >> 2, c_hi = a.end >> 2;
    const float fcount = (float)(a.end - a.start);
    const int ng = c_lo <= c_hi ? (c_hi - c_lo) / U : 0;
    const int c_end = c_lo + ng * U;
    for (int t = gw; t < a.tiles; t += a.waves) {
        const float4* tile = a.gal4 + (size_t)t * a.dp4 * 64 + lane;
        float acc[NH][QB];
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int q = 0; q < QB; ++q) acc[h][q] = 0.0f;
        if (c_lo > c_hi) {
            const float4 g = tile[(size_t)(a.start >> 2) * 64];
#pragma unroll
            for (int h = 0; h < NH; ++h) TileAcc<QB, METRIC, U>::masked(acc[h], g, qc[h], a.start >> 2, a.start, a.end);
        } else {
            if ((a.start & 3) != 0) {
                const float4 g = tile[(size_t)(c_lo - 1) * 64];
#pragma unroll
                for (int h = 0; h < NH; ++h) TileAcc<QB, METRIC, U>::masked(acc[h], g, qc[h], c_lo - 1, a.start, a.end);
            }
            const float4* p = tile + (size_t)c_lo * 64;
            for (int gi = 0; gi < ng; ++gi) {
                float4 g[U];
                ld_gallery_group<U>(g, p + (size_t)(gi * U) * 64, a.nt != 0);
#pragma unroll
                for (int h = 0; h < NH; ++h) TileAcc<QB, METRIC, U>::group(acc[h], g, qc[h], c_lo + gi * U);
            }
            for (int c = c_end; c < c_hi; ++c) {
                const float4 g = tile[(size_t)c * 64];
#pragma unroll
                for (int h = 0; h < NH; ++h) TileAcc<QB, METRIC, U>::masked(acc[h], g, qc[h], c, a.start, a.end);
            }
            if ((a.end & 3) != 0) {
                const float4 g = tile[(size_t)c_hi * 64];
#pragma unroll
                for (int h = 0; h < NH; ++h) TileAcc<QB, METRIC, U>::masked(acc[h], g, qc[h], c_hi, a.start, a.end);
            }
        }
        const int64_t row = (int64_t)t * kTileRows + lane;
        if (row < a.n) {
            const float sgr = a.sg[row];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
#pragma unroll
                for (int q = 0; q < QB; ++q) {
                    const size_t qy = ((size_t)blockIdx.y * NH + h) * QB + q;
                    float dist;
                    if constexpr (METRIC == kChi2Harm) dist = ((a.sq[qy] + sgr) - 4.0f * acc[h][q]) / fcount;
                    else dist = (0.693147181f * ((a.sq[qy] + sgr) - acc[h][q])) / fcount;
                    if (dist <= tau[h][q]) {
                        const int slot = atomicAdd(&a.counts[qy], 1);
                        if (slot < a.k) a.keys[qy * a.k + slot] = key_pack(dist, (uint32_t)(row + a.row_offset));
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Hand-scheduled L2 top-1 scan (the headline kernel): same tile streaming, same arithmetic and
// therefore the same bits as k_scan<QB, kL2, ..., kEpiTop1>, for feature ranges made of whole
// chunks (start % 4 == 0 && end % 4 == 0). The per-chunk body is one inline-asm block of 48
// packed-f32 VALU instructions for 8 queries (v_pk_add_f32 with an SGPR query pair and the lane's
// gallery value broadcast by op_sel, v_pk_mul_f32, v_pk_add_f32): packed f32 ops issue at the
// same rate as scalar ones on gfx950 (profiles/r01_ubench_valu_issue_rate.txt), i.e. two queries
// per issue slot, and four independent accumulator pairs are interleaved so no dependent pair is
// closer than four instructions (no hazard nops). IEEE-wise each half of a packed op is the plain
// f32 op, un-fused: sub, mul, add -- the reference's own three roundings.
// ---------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const f2 __attribute__((address_space(4)))* sf2_p;

#define FIR_SUB_LO(t, s, g) "v_pk_add_f32 %[" #t "], %[" #s "], %[" #g "] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define FIR_SUB_HI(t, s, g) "v_pk_add_f32 %[" #t "], %[" #s "], %[" #g "] op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]\n\t"
#define FIR_SQR(t) "v_pk_mul_f32 %[" #t "], %[" #t "], %[" #t "]\n\t"
#define FIR_ACC(a, t) "v_pk_add_f32 %[" #a "], %[" #a "], %[" #t "]\n\t"

// One pipeline unit: acc[0..3] (4 query pairs = 8 queries) += (q - g)^2 for TWO features held in
// the register pair g (lo = feature 2h, hi = feature 2h+1 of the chunk); s[jj*4+p] = queries
// (2p, 2p+1) of feature jj. 24 packed instructions, dependent ones >= 4 apart.
__device__ __forceinline__ void l2_half8(f2 (&a)[4], const f2 g, const f2 (&s)[8]) {
    f2 t0, t1, t2, t3, u0, u1, u2, u3;
    asm volatile(
        FIR_SUB_LO(t0, s0, g) FIR_SUB_LO(t1, s1, g) FIR_SUB_LO(t2, s2, g) FIR_SUB_LO(t3, s3, g)
        FIR_SUB_HI(u0, s4, g) FIR_SUB_HI(u1, s5, g) FIR_SUB_HI(u2, s6, g) FIR_SUB_HI(u3, s7, g)
        FIR_SQR(t0) FIR_SQR(t1) FIR_SQR(t2) FIR_SQR(t3)
        FIR_ACC(a0, t0) FIR_ACC(a1, t1) FIR_ACC(a2, t2) FIR_ACC(a3, t3)
        FIR_SQR(u0) FIR_SQR(u1) FIR_SQR(u2) FIR_SQR(u3)
        FIR_ACC(a0, u0) FIR_ACC(a1, u1) FIR_ACC(a2, u2) FIR_ACC(a3, u3)
        : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]),
          [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3),
          [u0] "=&v"(u0), [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3)
        : [g] "v"(g),
          [s0] "s"(s[0]), [s1] "s"(s[1]), [s2] "s"(s[2]), [s3] "s"(s[3]),
          [s4] "s"(s[4]), [s5] "s"(s[5]), [s6] "s"(s[6]), [s7] "s"(s[7]));
}

// Query values of unit (chunk c, half h, query block b): 2 features x 8 queries = 8 f2 (16 SGPRs).
template <int NB>
__device__ __forceinline__ void load_unit(f2 (&s)[8], sf2_p q2, int c, int h, int b) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int pp = 0; pp < 4; ++pp) s[jj * 4 + pp] = q2[((c * 4 + 2 * h + jj) * 4 * NB) + 4 * b + pp];
}

// One chunk (4 features) of one lane against 8*NB queries; the scalar loads run one unit ahead
// of the VALU work (cur = this unit's query values on entry, next unit's on exit).
template <int NB>
__device__ __forceinline__ void l2_chunk(f2 (&acc)[NB][4], const float4 g, f2 (&cur)[8], sf2_p q2, int c) {
    const f2 gp[2] = {f2{g.x, g.y}, f2{g.z, g.w}};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f2 nx[8];
            // the unit after (c, h, b); past the last chunk this reads the slack behind the query tile
            const int nb = b + 1 < NB ? b + 1 : 0;
            const int nh = b + 1 < NB ? h : (h + 1) & 1;
            const int nc = (b + 1 < NB || h == 0) ? c : c + 1;
            // SMEM returns out of order, so a wait for `cur` is lgkmcnt(0): take it BEFORE the next
            // unit's loads are issued (an empty asm that reads cur pins the wait here), then let
            // those loads fly under this unit's 24 VALU instructions.
            asm volatile("" ::"s"(cur[0]), "s"(cur[1]), "s"(cur[2]), "s"(cur[3]), "s"(cur[4]), "s"(cur[5]), "s"(cur[6]), "s"(cur[7]));
            __builtin_amdgcn_sched_barrier(0);
            load_unit<NB>(nx, q2, nc, nh, nb);
            l2_half8(acc[b], gp[h], cur);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) cur[i] = nx[i];
        }
    }
}

// NB blocks of 8 queries per pass (QB = 8 * NB). qt layout as everywhere: qt[k * QB + q]; the
// query tile must be followed by >= 64 readable floats (prefetch of the unit past the end).
template <int NB, int U, int WPS>
__global__ void __launch_bounds__(kBlock, WPS) k_scan_l2_fast(const ScanArgs a) {
    constexpr bool APPEND = false;     // the candidate-list form exists for the LDS-tile kernel only
    int32_t* counts = nullptr;
    constexpr int QB = 8 * NB;
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    // blockIdx.y = query tile: one launch runs several gallery passes back to back (no launch gaps, the
    // tail of one pass overlaps the head of the next; small galleries fill the chip with concurrent passes)
    sf2_p q2 = (sf2_p)(uintptr_t)(a.qt + (size_t)blockIdx.y * a.qt_stride);   // f2 index of (feature k, pair p of block b): k*4*NB + 4*b + p
    uint64_t* keys = a.keys + (size_t)blockIdx.y * QB;
    const int c_lo = a.start >> 2, c_hi = a.end >> 2;
    const float fcount = (float)(a.end - a.start);

    float best_d[QB];
    int32_t best_i[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) { best_d[q] = kNotFound; best_i[q] = -1; }

    for (int t = gw; t < a.tiles; t += a.waves) {
        const float4* p = a.gal4 + ((size_t)t * a.dp4 + c_lo) * 64 + lane;
        f2 acc[NB][4];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[b][i] = f2{0.0f, 0.0f};
        f2 cur[8];
        load_unit<NB>(cur, q2, c_lo, 0, 0);

        int c = c_lo;
        for (; c + U <= c_hi; c += U) {
            float4 g[U];
            ld_gallery_group<U>(g, p + (size_t)(c - c_lo) * 64, a.nt != 0);
#pragma unroll
            for (int u = 0; u < U; ++u) l2_chunk<NB>(acc, g[u], cur, q2, c + u);
        }
        for (; c < c_hi; ++c)      // (c_hi - c_lo) % U leftover chunks
            l2_chunk<NB>(acc, ld_gallery(p + (size_t)(c - c_lo) * 64, a.nt != 0), cur, q2, c);

        const int64_t row = (int64_t)t * kTileRows + lane;
        if (row < a.n) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float d0 = acc[b][i].x / fcount, d1 = acc[b][i].y / fcount;   // db_features.cpp:40
                    const int q0 = b * 8 + 2 * i;
                    if constexpr (APPEND) {
                        if (d0 <= best_d[q0]) {
                            const int slot = atomicAdd(&counts[q0], 1);
                            if (slot < a.k) keys[(size_t)q0 * a.k + slot] = key_pack(d0, (uint32_t)(row + a.row_offset));
                        }
                        if (d1 <= best_d[q0 + 1]) {
                            const int slot = atomicAdd(&counts[q0 + 1], 1);
                            if (slot < a.k) keys[(size_t)(q0 + 1) * a.k + slot] = key_pack(d1, (uint32_t)(row + a.row_offset));
                        }
                    } else {
                        if (d0 < best_d[q0]) { best_d[q0] = d0; best_i[q0] = (int32_t)row; }           // db_features.cpp:329-332
                        if (d1 < best_d[q0 + 1]) { best_d[q0 + 1] = d1; best_i[q0 + 1] = (int32_t)row; }
                    }
                }
        }
    }
    if constexpr (APPEND) return;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        uint64_t key = best_i[q] >= 0 ? key_pack(best_d[q], (uint32_t)((int64_t)best_i[q] + a.row_offset)) : kKeyNone;
        key = wave_min_u64(key);
        if (lane == 0 && key != kKeyNone) {
            const uint64_t cur_key = __hip_atomic_load(keys + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (key < cur_key) atomicMin((unsigned long long*)(keys + q), (unsigned long long)key);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Same kernel with the query tile staged in LDS. Why: the SGPR form above fetches 16 query
// dwords per 24 VALU instructions through the scalar cache; an 8-query tile at d = 512 is 16 KiB
// -- the whole scalar cache -- and every wave walks it at its own phase, so the s_loads miss to L2
// (~800 cycles each, measured: 0.7 us per chunk per wave, profiles/r01_sweep_notes.md) and SMEM
// returns out of order, so only one batch can be in flight per wave: the wave becomes latency
// bound. LDS reads are in-order, counted (lgkmcnt(N)) and ~100 cycles: every lane reads the same
// address (a broadcast, no bank conflict), 4 x ds_read_b128 per unit, issued one unit ahead.
// ---------------------------------------------------------------------------------------------
// One feature of one lane against 8 queries: acc[0..3] += (q - g)^2, g = lo or hi half of the
// register pair, s[p] = queries (2p, 2p+1). 12 packed instructions, dependent ones 4 apart.
template <int HI>
__device__ __forceinline__ void l2_feat8_v(f2 (&a)[4], const f2 g, const f2 (&s)[4]) {
    f2 t0, t1, t2, t3;
    if constexpr (HI == 0) {
        asm volatile(FIR_SUB_LO(t0, s0, g) FIR_SUB_LO(t1, s1, g) FIR_SUB_LO(t2, s2, g) FIR_SUB_LO(t3, s3, g)
                     FIR_SQR(t0) FIR_SQR(t1) FIR_SQR(t2) FIR_SQR(t3)
                     FIR_ACC(a0, t0) FIR_ACC(a1, t1) FIR_ACC(a2, t2) FIR_ACC(a3, t3)
                     : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]),
                       [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [g] "v"(g), [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [s3] "v"(s[3]));
    } else {
        asm volatile(FIR_SUB_HI(t0, s0, g) FIR_SUB_HI(t1, s1, g) FIR_SUB_HI(t2, s2, g) FIR_SUB_HI(t3, s3, g)
                     FIR_SQR(t0) FIR_SQR(t1) FIR_SQR(t2) FIR_SQR(t3)
                     FIR_ACC(a0, t0) FIR_ACC(a1, t1) FIR_ACC(a2, t2) FIR_ACC(a3, t3)
                     : [a0] "+v"(a[0]), [a1] "+v"(a[1]), [a2] "+v"(a[2]), [a3] "+v"(a[3]),
                       [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [g] "v"(g), [s0] "v"(s[0]), [s1] "v"(s[1]), [s2] "v"(s[2]), [s3] "v"(s[3]));
    }
}

// Query values of unit (feature k, block b) from the LDS copy of the tile (float4 view:
// index of (feature k, queries 4i..4i+3) = k * 2 * NB + i): 2 x ds_read_b128, a broadcast.
template <int NB>
__device__ __forceinline__ void lds_unit(f2 (&s)[4], const float4* lq, int k, int b) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float4 v = lq[k * 2 * NB + 2 * b + i];
        s[2 * i] = f2{v.x, v.y};
        s[2 * i + 1] = f2{v.z, v.w};
    }
}

// One chunk (4 features) of one lane against 8*NB queries; the LDS reads run one unit (one
// feature x 8 queries) ahead of the VALU work: cur = this unit's query values on entry, the next
// unit's on exit (past the last feature: the zeroed slack chunk).
// R rows per lane (R consecutive tiles per wave and step): every unit's two LDS reads then serve R rows -- at 16 queries per read of
// the gallery the CU's LDS pipe is two thirds busy when the vector pipes run at their full rate (profiles/r04_scan16_forms.txt).
template <int NB, int R = 1>
__device__ __forceinline__ void l2_chunk_lds(f2 (&acc)[R][NB][4], const float4 (&g)[R], f2 (&cur)[4], const float4* lq, int c) {
    f2 gp[R][2];
#pragma unroll
    for (int r = 0; r < R; ++r) { gp[r][0] = f2{g[r].x, g[r].y}; gp[r][1] = f2{g[r].z, g[r].w}; }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            f2 nx[4];
            const int nb = b + 1 < NB ? b + 1 : 0;
            const int nk = c * 4 + j + (b + 1 < NB ? 0 : 1);
            lds_unit<NB>(nx, lq, nk, nb);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (j & 1) l2_feat8_v<1>(acc[r][b], gp[r][j >> 1], cur);
                else l2_feat8_v<0>(acc[r][b], gp[r][j >> 1], cur);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) cur[i] = nx[i];
        }
    }
}

// Dynamic LDS: (dp4 + 1) * 4 * QB floats (the query tile + one zero chunk of slack).
// APPEND = true: instead of a running first minimum, every row whose distance is <= tau[query] is appended (packed key) to
// that query's candidate list: the distances are the exact ones, so the K nearest rows are the K smallest keys of a list
// whose threshold came from a row sample (fir_capi.hip, topk_dev) -- the top-K scan at the speed of the top-1 scan.
template <int NB, int U, int WPS, bool APPEND = false, int R = 1>
__global__ void __launch_bounds__(kBlock, WPS) k_scan_l2_lds(const ScanArgs a) {
    constexpr int QB = 8 * NB;
    extern __shared__ __attribute__((aligned(16))) float4 lq[];
    const int lane = threadIdx.x & 63;
    const int gw = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    const int c_lo = a.start >> 2, c_hi = a.end >> 2;
    const float fcount = (float)(a.end - a.start);
    {
        // blockIdx.y = query tile (see k_scan_l2_fast)
        const float4* src = reinterpret_cast<const float4*>(a.qt + (size_t)blockIdx.y * a.qt_stride);
        const int n4 = a.dp4 * QB;                       // float4s in the tile: dp4*4 features * QB / 4
        for (int i = threadIdx.x; i < n4 + QB; i += kBlock) lq[i] = i < n4 ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        __syncthreads();
    }
    // LDS base held in a VGPR the compiler cannot prove uniform: the reads then use one per-group
    // address register + immediate offsets (ds_read_b128 v, vaddr offset:N) instead of one
    // s_add + v_mov per read. All lanes still read the same address (hardware broadcast).
    int zero_v;
    asm volatile("v_mov_b32 %0, 0" : "=v"(zero_v));
    const float4* lqv = lq + zero_v;
    uint64_t* keys = a.keys + (size_t)blockIdx.y * QB * (APPEND ? a.k : 1);

    float best_d[QB];      // APPEND: the thresholds tau
    int32_t best_i[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        best_d[q] = APPEND ? a.tau[(size_t)blockIdx.y * QB + q] : kNotFound;
        best_i[q] = -1;
    }
    int32_t* counts = APPEND ? a.counts + (size_t)blockIdx.y * QB : nullptr;

    // a wave's step: R consecutive tiles, a lane owns row `lane` of each (ascending: the running first minimum stays the reference's);
    // the last step may repeat the last tile (its rows are not looked at twice: the epilogue checks the tile index)
    const int tsteps = (a.tiles + R - 1) / R;
    for (int ts = gw; ts < tsteps; ts += a.waves) {
        const float4* p[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = ts * R + r < a.tiles ? ts * R + r : a.tiles - 1;
#ifdef FIR_DEBUG_TILEMOD   // timing experiments only: every tile reads one of the first few (cache-resident stream)
            p[r] = a.gal4 + ((size_t)(t % FIR_DEBUG_TILEMOD) * a.dp4 + c_lo) * 64 + lane;
#else
            p[r] = a.gal4 + ((size_t)t * a.dp4 + c_lo) * 64 + lane;
#endif
        }
        f2 acc[R][NB][4];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[r][b][i] = f2{0.0f, 0.0f};
        f2 cur[4];
        lds_unit<NB>(cur, lqv, c_lo * 4, 0);

        int c = c_lo;
#if FIR_PIPE == 0
        for (; c + U <= c_hi; c += U) {
            float4 g[R][U];
#pragma unroll
            for (int r = 0; r < R; ++r) ld_gallery_group<U>(g[r], p[r] + (size_t)(c - c_lo) * 64, a.nt != 0);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                float4 gu[R];
#pragma unroll
                for (int r = 0; r < R; ++r) gu[r] = g[r][u];
                l2_chunk_lds<NB, R>(acc, gu, cur, lqv, c + u);
            }
        }
#else
        // register double buffer: the next group's loads are in flight while this group is consumed
        {
            const int ng = (c_hi - c_lo) / U;
            float4 g[R][U];
            if (ng > 0) {
#pragma unroll
                for (int r = 0; r < R; ++r) ld_gallery_group<U>(g[r], p[r], a.nt != 0);
            }
            for (int gi = 0; gi < ng; ++gi, c += U) {
                float4 nxg[R][U];
                const int gn = gi + 1 < ng ? gi + 1 : gi;   // the last group re-reads itself (L2 hit; keeps the loop branch-free)
#pragma unroll
                for (int r = 0; r < R; ++r) ld_gallery_group<U>(nxg[r], p[r] + (size_t)(gn * U) * 64, a.nt != 0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    float4 gu[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) gu[r] = g[r][u];
                    l2_chunk_lds<NB, R>(acc, gu, cur, lqv, c + u);
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int u = 0; u < U; ++u) g[r][u] = nxg[r][u];
            }
        }
#endif
        for (; c < c_hi; ++c) {
            float4 gu[R];
#pragma unroll
            for (int r = 0; r < R; ++r) gu[r] = ld_gallery(p[r] + (size_t)(c - c_lo) * 64, a.nt != 0);
            l2_chunk_lds<NB, R>(acc, gu, cur, lqv, c);
        }

#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int t = ts * R + r;
            const int64_t row = (int64_t)t * kTileRows + lane;
            if (t < a.tiles && row < a.n) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const float d0 = acc[r][b][i].x / fcount, d1 = acc[r][b][i].y / fcount;   // db_features.cpp:40
                        const int q0 = b * 8 + 2 * i;
                        if constexpr (APPEND) {
                            if (d0 <= best_d[q0]) {
                                const int slot = atomicAdd(&counts[q0], 1);
                                if (slot < a.k) keys[(size_t)q0 * a.k + slot] = key_pack(d0, (uint32_t)(row + a.row_offset));
                            }
                            if (d1 <= best_d[q0 + 1]) {
                                const int slot = atomicAdd(&counts[q0 + 1], 1);
                                if (slot < a.k) keys[(size_t)(q0 + 1) * a.k + slot] = key_pack(d1, (uint32_t)(row + a.row_offset));
                            }
                        } else {
                            if (d0 < best_d[q0]) { best_d[q0] = d0; best_i[q0] = (int32_t)row; }           // db_features.cpp:329-332
                            if (d1 < best_d[q0 + 1]) { best_d[q0 + 1] = d1; best_i[q0 + 1] = (int32_t)row; }
                        }
                    }
            }
        }
    }
    if constexpr (APPEND) return;
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        uint64_t key = best_i[q] >= 0 ? key_pack(best_d[q], (uint32_t)((int64_t)best_i[q] + a.row_offset)) : kKeyNone;
        key = wave_min_u64(key);
        if (lane == 0 && key != kKeyNone) {
            const uint64_t cur_key = __hip_atomic_load(keys + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (key < cur_key) atomicMin((unsigned long long*)(keys + q), (unsigned long long)key);
        }
    }
}

// Final merge of the per-wave top-K partials: one block per query. K rounds of
// "smallest key greater than the previous winner" over waves*K candidates.
__global__ void __launch_bounds__(kBlock) k_topk_merge(const uint64_t* part, int waves, int QB, int q0, int nq, int K,
                                                        uint64_t* out /* [nq_total][K] */) {
    __shared__ uint64_t red[kBlock / 64];
    const int q = blockIdx.x;   // query inside the tile
    if (q >= nq) return;
    uint64_t prev = 0;
    bool first = true;
    for (int r = 0; r < K; ++r) {
        uint64_t cand = kKeyNone;
        for (int i = threadIdx.x; i < waves * K; i += kBlock) {
            const int w = i / K, j = i - w * K;
            const uint64_t v = part[((size_t)w * QB + q) * K + j];
            if ((first || v > prev) && v < cand) cand = v;
        }
        cand = wave_min_u64(cand);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cand;
        __syncthreads();
        uint64_t m = red[0];
#pragma unroll
        for (int i = 1; i < kBlock / 64; ++i) m = red[i] < m ? red[i] : m;
        __syncthreads();
        if (threadIdx.x == 0) out[(size_t)(q0 + q) * K + r] = m;
        prev = m;
        first = false;
    }
}

// Thresholds of the append form. gkeys[i][q] = nearest row of query q inside the i-th of K DISJOINT row groups of a
// sample (top-1 scans): tau[q] = the largest of those K distances, so at least K rows (one per group) are <= tau[q].
// A group without any row below 100000 raises the flag (the caller then uses the register-list scan). Padding queries
// [nq, nq_pad) get a threshold nothing reaches.
__global__ void k_topk_tau(const uint64_t* __restrict__ gkeys, int nq, int nq_pad, int K, float* __restrict__ tau, int32_t* __restrict__ counts,
                           int32_t* __restrict__ flag, float scale = 1.0f, int stride = 0) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= nq_pad) return;
    if (stride == 0) stride = nq;                                    // keys of sample group i: gkeys[i * stride + q]
    counts[q] = 0;
    if (q >= nq) { tau[q] = -__builtin_huge_valf(); return; }       // (-inf: no distance is at or below it -- a KL value can round to a small negative number, -1 could not tell)
    uint64_t worst = 0;
    for (int i = 0; i < K; ++i) {
        const uint64_t key = gkeys[(size_t)i * stride + q];
        worst = key > worst ? key : worst;
    }
    if (worst == kKeyNone) { tau[q] = -__builtin_huge_valf(); atomicOr(flag, 1); return; }
    // scale > 1: the append scan that follows uses a cheaper metric whose value is within (scale - 1) of the reference's
    tau[q] = f32_from_orderable((uint32_t)(worst >> 32)) * scale;
}
// The K smallest keys of each query's candidate list (count <= cap entries; more raises the flag). One block per query.
__global__ void __launch_bounds__(kBlock) k_topk_select(const uint64_t* __restrict__ lists, const int32_t* __restrict__ counts, int cap, int K,
                                                         uint64_t* __restrict__ out, int32_t* __restrict__ flag) {
    __shared__ uint64_t red[kBlock / 64];
    const int q = blockIdx.x;
    const int cnt = counts[q];
    if (cnt > cap || cnt < K) { if (threadIdx.x == 0) atomicOr(flag, 1); return; }
    const uint64_t* l = lists + (size_t)q * cap;
    uint64_t prev = 0;
    bool first = true;
    for (int r = 0; r < K; ++r) {
        uint64_t cand = kKeyNone;
        for (int i = threadIdx.x; i < cnt; i += kBlock) {
            const uint64_t v = l[i];
            if ((first || v > prev) && v < cand) cand = v;
        }
        cand = wave_min_u64(cand);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cand;
        __syncthreads();
        uint64_t m = red[0];
#pragma unroll
        for (int i = 1; i < kBlock / 64; ++i) m = red[i] < m ? red[i] : m;
        __syncthreads();
        if (threadIdx.x == 0) out[(size_t)q * K + r] = m;
        prev = m;
        first = false;
    }
}

// Exact re-rank of a candidate list whose keys were written by a NOMINATION scan (kChi2Approx): every entry's distance is
// recomputed with the reference's arithmetic (accum<METRIC>, ascending feature order, one IEEE division by the count) and the
// key rewritten in place; k_topk_select then picks the K smallest. One block per query, one thread per entry; the query sits in LDS.
template <int METRIC>
__global__ void __launch_bounds__(kBlock) k_list_rerank(uint64_t* __restrict__ lists, const int32_t* __restrict__ counts, int cap,
                                                         const float4* __restrict__ gal4, int dp4, int64_t n, int64_t row_offset,
                                                         const float* __restrict__ queries, int d, int start, int end) {
    extern __shared__ float lq_rr[];
    const int q = blockIdx.x;
    for (int k = threadIdx.x; k < d; k += kBlock) lq_rr[k] = queries[(size_t)q * d + k];
    __syncthreads();
    const int cnt = counts[q] < cap ? counts[q] : cap;
    const float fcount = (float)(end - start);
    uint64_t* l = lists + (size_t)q * cap;
    for (int i = threadIdx.x; i < cnt; i += kBlock) {
        const int64_t row = (int64_t)(uint32_t)(l[i] & 0xFFFFFFFFull) - row_offset;
        uint64_t key = kKeyNone;
        if (row >= 0 && row < n) {
            const float4* gr = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
            float acc = 0.0f;
            // (a lane gathers its row 16 bytes per 4 features: eight loads in flight, the sum itself stays sequential)
            const int c_first = start >> 2, c_last = (end - 1) >> 2;
            for (int c0 = c_first; c0 <= c_last; c0 += 8) {
                float4 g8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g8[u] = gr[(size_t)(c0 + u <= c_last ? c0 + u : c_last) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = c0 + u;
                    if (c <= c_last) {
                        const float gv[4] = {g8[u].x, g8[u].y, g8[u].z, g8[u].w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int k = c * 4 + j;
                            if (k >= start && k < end) acc = accum<METRIC>(acc, lq_rr[k], gv[j]);
                        }
                    }
                }
            }
            const float dist = acc / fcount;
            if (dist < kNotFound) key = key_pack(dist, (uint32_t)(row + row_offset));
        }
        l[i] = key;
    }
}

// rows[n][d] row-major  ->  tiled layout (see top of file). One thread per output float4.
// row0 = first row of this slab (multiple of 64); rows beyond n and features beyond d are zero.
// range[0] is set when a value outside in_plain_range() goes by (the chi-square / KL scans then keep the full division).
__global__ void __launch_bounds__(kBlock) k_retile(const float* __restrict__ rows, int64_t slab_rows, int64_t row0,
                                                    int64_t n, int d, int dp4, float4* __restrict__ gal4,
                                                    int32_t* __restrict__ range) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;   // float4 index inside the slab's tiles
    const int64_t slab_tiles = (slab_rows + kTileRows - 1) / kTileRows;
    if (o >= slab_tiles * dp4 * 64) return;
    const int r = (int)(o & 63);
    const int64_t tc = o >> 6;
    const int c = (int)(tc % dp4);
    const int64_t tl = tc / dp4;
    const int64_t lrow = tl * kTileRows + r;       // row inside the slab
    const int64_t grow = row0 + lrow;              // row inside the gallery
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (lrow < slab_rows && grow < n) {
        const float* src = rows + lrow * d + (int64_t)c * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c * 4 + j < d) v[j] = src[j];
    }
    if (range && !(in_plain_range(v[0]) && in_plain_range(v[1]) && in_plain_range(v[2]) && in_plain_range(v[3]))) range[0] = 1;
    gal4[((row0 / kTileRows + tl) * dp4 + c) * 64 + r] = make_float4(v[0], v[1], v[2], v[3]);
}

// queries[nq][d] row-major -> tiles of QB queries, transposed: qt[tile][k][QB], k < dp4*4.
// Queries past nq and features past d are zero. keys[0..nkeys) (may be NULL) are preset to "no row yet" on the way:
// the top-1 scans that follow only ever lower them.
// range[1] = serial when a query value outside in_plain_range() goes by (serial numbers the transpositions of a handle).
// recip == 1 (kChi2Harm): the tile holds 1 / value (2^60 for 0 and for the padding), the range check still sees the value;
// recip == 2 (kKLEnt): value + 2^-100 (the value itself unless it is 0: plain-range values are 0 or >= 2^-26).
__global__ void __launch_bounds__(kBlock) k_transpose_queries(const float* __restrict__ q, int nq, int d, int dp4, int QB,
                                                               float* __restrict__ qt, uint64_t* __restrict__ keys, int nkeys,
                                                               int32_t* __restrict__ range = nullptr, int serial = 0, int recip = 0) {
    const int kk = dp4 * 4;
    const int64_t total = (int64_t)((nq + QB - 1) / QB) * kk * QB;
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (keys && o < nkeys) keys[o] = kKeyNone;
    if (o >= total) return;
    // consecutive threads read consecutive features of one query (the source may be pinned host memory read over
    // PCIe: coalesced reads matter there); the strided side is the write into device memory
    const int k = (int)(o % kk);
    const int64_t r = o / kk;
    const int qi = (int)(r % QB);
    const int tile = (int)(r / QB);
    const int qq = tile * QB + qi;
    const float v = (qq < nq && k < d) ? q[(int64_t)qq * d + k] : 0.0f;
    if (range && !in_plain_range(v)) range[1] = serial;
    qt[((int64_t)tile * kk + k) * QB + qi] = recip == 1 ? fminf(1.0f / v, 0x1p60f) : recip == 2 ? v + 0x1p-100f : v;
}

// kChi2Harm: sg[row] = sum of the row's values over features [start, end) (one lane per row, tiled layout), smax[0] = their
// largest (float bits, atomicMax: the sums are non-negative for the plain-range galleries the metric is used on)
__global__ void __launch_bounds__(kBlock) k_row_sums(const float4* __restrict__ gal4, int64_t n, int dp4, int start, int end, float* __restrict__ sg,
                                                      unsigned int* __restrict__ smax) {
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float s = 0.0f;
    if (row < n) {
        const float4* p = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
        for (int c = start >> 2; c <= (end - 1) >> 2; ++c) {
            const float4 g = p[(size_t)c * 64];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = c * 4 + j;
                if (k >= start && k < end) s += gv[j];
            }
        }
        sg[row] = s;
    }
    float m = s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(smax, __float_as_uint(m));
}
// kKLEnt: sg[row] = sum over [start, end) of r log2 r + r (0 for r = 0; added up in double, stored as float), smax[0] = the largest
// plain row sum (float bits, atomicMax) -- what the nomination's error bound is relative to, as for kChi2Harm
__global__ void __launch_bounds__(kBlock) k_row_entropy(const float4* __restrict__ gal4, int64_t n, int dp4, int start, int end, float* __restrict__ sg,
                                                         unsigned int* __restrict__ smax) {
    const int64_t row = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    float s = 0.0f;
    if (row < n) {
        const float4* p = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
        double e = 0.0;
        for (int c = start >> 2; c <= (end - 1) >> 2; ++c) {
            const float4 g = p[(size_t)c * 64];
            const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = c * 4 + j;
                if (k >= start && k < end && gv[j] > 0.0f) { s += gv[j]; e += (double)gv[j] * log2((double)gv[j]) + (double)gv[j]; }
            }
        }
        sg[row] = (float)e;
    }
    float m = s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.0f) atomicMax(smax, __float_as_uint(m));
}
// ... and the same for the queries of a call; tau[q] += 1.5 * B, B = coef * (sum(l) + smax) the bound on |entropy form - reference|
__global__ void __launch_bounds__(64) k_query_entropy_widen(const float* __restrict__ q, int nq, int d, int start, int end, float* __restrict__ sq,
                                                            float* __restrict__ tau, const unsigned int* __restrict__ smax, float coef) {
    const int qi = blockIdx.x;
    float s = 0.0f;
    double e = 0.0;
    if (qi < nq)
        for (int k = start + threadIdx.x; k < end; k += 64) {
            const float v = q[(size_t)qi * d + k];
            if (v > 0.0f) { s += v; e += (double)v * log2((double)v) + (double)v; }
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { s += __shfl_xor(s, off, 64); e += __shfl_xor(e, off, 64); }
    if (threadIdx.x == 0) {
        sq[qi] = (float)e;
        if (qi < nq && tau[qi] > -__builtin_huge_valf()) tau[qi] += 1.5f * coef * (s + __uint_as_float(smax[0]));    // (every sampled threshold is widened, a slightly negative one too: ADVICE r3)
    }
}
// sq[q] = sum of query q over [start, end); tau[q] += 1.5 * B, B = coef * (sq[q] + smax) the bound on |harmonic form - reference|
// (fir_capi.hip, topk_lists_dev): every row whose reference distance is <= the unwidened threshold passes the widened one
__global__ void __launch_bounds__(64) k_query_sums_widen(const float* __restrict__ q, int nq, int d, int start, int end, float* __restrict__ sq,
                                                         float* __restrict__ tau, const unsigned int* __restrict__ smax, float coef) {
    const int qi = blockIdx.x;
    float s = 0.0f;
    if (qi < nq)
        for (int k = start + threadIdx.x; k < end; k += 64) s += q[(size_t)qi * d + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) {
        sq[qi] = s;
        if (qi < nq && tau[qi] > -__builtin_huge_valf()) tau[qi] += 1.5f * coef * (s + __uint_as_float(smax[0]));    // (every sampled threshold is widened, a slightly negative one too: ADVICE r3)
    }
}

// feature_distance for one pair (db_features.cpp:22-42): one lane, sequential, exact order.
template <int METRIC>
__global__ void k_pair_distance(const float* __restrict__ l, const float* __restrict__ r, int start, int end, float* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float acc = 0.0f;
    for (int i = start; i < end; ++i) acc = accum<METRIC>(acc, l[i], r[i]);
    *out = acc / (float)(end - start);
}

__global__ void __launch_bounds__(kBlock) k_classes_of(const int32_t* __restrict__ cls, int64_t n, int64_t row_offset,
                                                        const int32_t* __restrict__ idx, int m, int32_t* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= m) return;
    const int64_t l = (int64_t)idx[i] - row_offset;
    out[i] = (idx[i] >= 0 && l >= 0 && l < n) ? cls[l] : -1;
}

}  // namespace fir
