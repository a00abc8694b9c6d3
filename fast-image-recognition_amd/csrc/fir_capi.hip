// fir_capi.hip -- the C ABI (include/fir_amd.h) over the gfx950 kernels in fir_kernels.h.
//
// Host side of the drop-in boundary: owns device buffers, streams and launch geometry.
// Nothing here computes a distance on the host; without a HIP device every entry point fails.
#include "fir_kernels.h"

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"

using namespace fir;

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define FIR_HIP(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,     \
                                          "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

#ifndef FIR_U
#define FIR_U 8
#endif
#ifndef FIR_WPS
#define FIR_WPS 4
#endif
constexpr int kU = FIR_U;    // float4 chunks per lane per load group (FIR_U KiB per wave in flight per group)
constexpr int kWps = FIR_WPS;     // waves per SIMD the scan kernels are register-budgeted for (<= 128 VGPRs)
// The plain-range chi-square / KL kernels pair their operands for v_pk_fma_f32 and want more registers (with 128 the
// chi-square loop spills, and every reload waits for the gallery loads in flight); pick_waves never asks for more than 3.
constexpr int kWpsPlain = 3;
constexpr int wps_of(int metric) { return metric >= kChi2InRange ? kWpsPlain : kWps; }
constexpr int kKMax = 8;     // per-lane candidate list length of the top-K scan

}  // namespace

struct fir_gallery {
    int device = 0;
    int cus = 0;
    int64_t n = 0;
    int d = 0, dp4 = 0;
    int64_t tiles = 0;
    int metric = FIR_METRIC_L2;
    int64_t row_offset = 0;
    float4* gal4 = nullptr;
    int32_t* cls = nullptr;
    hipStream_t stream = nullptr;
    // chi-square / KL: range[0] = a gallery value outside in_plain_range() was uploaded, range[1] = serial of the last query
    // transposition that met one (fir_common.h); the scans read both and pick their division sequence on the device
    int32_t* range = nullptr;
    int q_serial = 0;
    bool gallery_plain = false;   // host copy of range[0] == 0, read once after the upload (chi-square nomination, topk_lists_dev)
    uint64_t* one_keys = nullptr;   // one-query calls on small galleries (top1_one_query): device key + completion counter, armed once
    int32_t* one_done = nullptr;    // and re-armed by the kernel itself
    uint64_t one_ticket = 0;        // number of such calls so far: the word the host waits for

    // workspaces (grown on demand, never inside a *_dev call once large enough)
    float* qt = nullptr;      size_t qt_cap = 0;      // transposed query tiles
    float* dq = nullptr;      size_t dq_cap = 0;      // staged host queries
    uint64_t* dkeys = nullptr; size_t dkeys_cap = 0;  // keys for the host-pointer API
    uint64_t* part = nullptr; size_t part_cap = 0;    // top-K per-wave partials
    float* dout = nullptr;    size_t dout_cap = 0;    // range distances for the host-pointer API
    int32_t* didx = nullptr;  size_t didx_cap = 0;
    // pinned, device-visible host staging of the small host-pointer calls: queries go in, packed keys come out, with
    // no copy engine in between (the kernels read / write it over PCIe) and one stream synchronisation per call
    void* pin = nullptr;
    uint64_t counters[4] = {};  // fir_gallery_next_counter_
    void* scratch[24] = {};   size_t scratch_cap[24] = {};   // fir_gallery_scratch_ (classifier entry points in the other translation units)

    int qpp = 0;              // queries per gallery pass; 0 = automatic (effective_qpp)
    int waves_req = 0;        // 0 = automatic
    int max_waves = 0;        // upper bound over all scan kernels (8 blocks per CU)
    int last_waves = 0;       // waves of the most recent scan launch
    // L2 whole-range batches of at least this many queries go through fir_gemm_* (same keys): -1 = automatic
    // (kAutoMfmaQueries queries against at least kAutoMfmaRows rows), 0 = never, > 0 = the caller's threshold
    int large_batch_min = -1;
    // the matrix-core states of the last TWO feature prefixes [0, end) asked for: the reference's harness alternates "BF, 64" and
    // "BF, 256" (ImageTesting.cpp:526-529); one slot would rebuild fragments and scratch on every call
    struct PrefixSlot { fir_gemm* m = nullptr; int end = 0; uint64_t used = 0; } gemm_prefix[2];
    uint64_t prefix_clock = 0;
    float* rowsum = nullptr; size_t rowsum_cap = 0;    // chi-square nomination (kChi2Harm): per-row sums over [rs_start, rs_end) + their maximum (rowsum[n])
    int rs_start = -1, rs_end = -1, rs_metric = -1;   // (what g->rowsum holds: row sums for chi-square, row entropies for KL)
    int warm_left = 0;                  // fir_dispatch_info::warmup_calls_left of the most recent call
    int shadow_mode = FIR_SHADOW_ALL;   // which copies of the gallery the automatic dispatch may keep next to the tiled f32 rows (fir_gallery_set_shadow_copies)
    int small_hits = 0, few_hits = 0;   // automatic mode: calls so far that would have profited from a matrix-core state not built yet (see ensure_gemm)
    bool gemm_failed = false; // automatic mode: the matrix-core path could not be set up for this shape (rows too long): scan
    fir_gemm* gemm = nullptr; // created on first use
    fir_dispatch_info last{}; // dominant kernel of the most recent search
    int call_launches = 0;    // scan launches of the current call (note_dispatch)
    bool quiet = false;       // scans on behalf of another path (the matrix-core path's uncertified queries): not recorded, not timed
    int sample_groups = 0; int64_t sample_group_stride = 0;   // run_pass (generic k_scan, top-1): ScanArgs::groups / group_stride
    int64_t tiles_limit = 0, tile_begin = 0;  // tiles_limit > 0: scans cover tiles [tile_begin, tile_begin + tiles_limit) only (row samples of the top-K threshold)
    int max_tiles_per_launch = 64;   // query tiles (gallery passes) folded into one launch of the hand-scheduled kernels

    struct Occ { const void* fn; size_t lds; int waves; };
    std::vector<Occ> occ;     // resident-wave capacity per scan kernel

    bool profiling = false;
    std::vector<hipEvent_t> ev;   // pairs
    size_t ev_used = 0;
    double last_bytes = 0.0;
};

namespace {

template <typename T>
int grow(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return FIR_OK;
    if (p) FIR_HIP(hipFree(p));
    p = nullptr; cap = 0;
    size_t want = std::max(need, (size_t)4096);
    FIR_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return FIR_OK;
}

typedef void (*scan_fn)(const ScanArgs);

template <int EPI>
scan_fn pick_kernel(int qb, int metric) {
#define FIR_CASE(QB, M) if (qb == QB && metric == M) return (scan_fn)k_scan<QB, M, kU, EPI, kKMax, wps_of(M)>;
#ifdef FIR_MINIMAL   // experiment builds: only the L2 top-1 kernels
    if constexpr (EPI == kEpiTop1) { FIR_CASE(8, 0) FIR_CASE(4, 0) FIR_CASE(2, 0) FIR_CASE(1, 0) }
    return nullptr;
#endif
    FIR_CASE(1, 0) FIR_CASE(2, 0) FIR_CASE(4, 0)
    FIR_CASE(1, 1) FIR_CASE(2, 1) FIR_CASE(4, 1)
    FIR_CASE(1, 2) FIR_CASE(2, 2) FIR_CASE(4, 2)
    FIR_CASE(1, 3) FIR_CASE(2, 3) FIR_CASE(4, 3)      // 3, 4: the plain-range forms of 1, 2 (fir_common.h)
    FIR_CASE(1, 4) FIR_CASE(2, 4) FIR_CASE(4, 4)
    if constexpr (EPI != kEpiTopK) {   // the top-K scan keeps 2*kKMax registers per query: 4 queries at most
        FIR_CASE(8, 0) FIR_CASE(8, 1) FIR_CASE(8, 2) FIR_CASE(8, 3) FIR_CASE(8, 4)
    }
#undef FIR_CASE
    return nullptr;
}

#ifndef FIR_FAST
#define FIR_FAST 1
#endif
#ifndef FIR_FAST_U
#define FIR_FAST_U 8
#endif
#ifndef FIR_FAST_WPS
#define FIR_FAST_WPS 4
#endif
#ifndef FIR_FAST_MODE
#define FIR_FAST_MODE 2   // 1: query tile through the scalar cache (SGPR operands), 2: staged in LDS
#endif
// The hand-scheduled L2 top-1 kernels cover whole-chunk feature ranges with 8 or 16 queries.
// *lds_bytes receives the dynamic LDS size the kernel must be launched with.
constexpr size_t kMaxQueryTileLds = 144 * 1024;   // of the CU's 160 KiB; one workgroup per CU above 80 KiB
scan_fn pick_fast(int epi, int qb, int metric, int start, int end, int dp4, size_t* lds_bytes) {
    if (lds_bytes) *lds_bytes = 0;
#if FIR_FAST
    if (epi != kEpiTop1 || metric != kL2 || (start & 3) || (end & 3)) return nullptr;
    if (qb != 8 && qb != 16) return nullptr;
#if FIR_FAST_MODE == 2
    const size_t need = (size_t)(dp4 + 1) * qb * 16;     // query tile + one zero chunk of slack
    if (need <= kMaxQueryTileLds) {                        // above 64 KiB the launch opts in (run_pass)
        if (lds_bytes) *lds_bytes = need;
        if (qb == 8) return (scan_fn)k_scan_l2_lds<1, FIR_FAST_U, FIR_FAST_WPS>;
        // 16 queries per read: two rows per lane (fir_kernels.h, l2_chunk_lds). FIR_SCAN16_FORM (experiments): 0 = one row per lane (rounds 1-3)
        static const int form16 = fir_knob_("FIR_SCAN16_FORM") ? std::atoi(fir_knob_("FIR_SCAN16_FORM")) : 3;
        switch (form16) {
            // (profiles/r04_scan16_forms.txt: 1M x 512 32.3 k q/s -> 33.2-33.5 k with two rows per lane whatever U / waves per SIMD; the vector
            // pipes are the bound at the clock the chip holds, not the LDS reads)
            case 0: return (scan_fn)k_scan_l2_lds<2, FIR_FAST_U, FIR_FAST_WPS>;
            case 1: return (scan_fn)k_scan_l2_lds<2, 4, 4, false, 2>;
            case 2: return (scan_fn)k_scan_l2_lds<2, 8, 3, false, 2>;
            case 4: return (scan_fn)k_scan_l2_lds<2, 2, 4, false, 2>;
            default: return (scan_fn)k_scan_l2_lds<2, 4, 3, false, 2>;
        }
    }
#endif
    if (qb == 8) return (scan_fn)k_scan_l2_fast<1, FIR_FAST_U, FIR_FAST_WPS>;
    return (scan_fn)k_scan_l2_fast<2, FIR_FAST_U, FIR_FAST_WPS>;
#endif
    return nullptr;
}

// Few tiles (a gallery of a few thousand rows): every wave owns a whole tile and streams it alone, so what limits a
// single-query call is how many gallery loads the wave keeps in flight. One- and two-query tiles have the registers
// for 16 chunks (16 KiB per wave) per load group.
constexpr int kUDeep = 16;
scan_fn pick_deep(int epi, int qb, int metric, int dp4, size_t* lds_bytes) {
    const size_t need = (size_t)dp4 * 4 * qb * sizeof(float);
    if (need > 64 * 1024) return nullptr;
#ifndef FIR_MINIMAL
#define FIR_DEEP(QB, M, E) if (epi == E && qb == QB && metric == M) { *lds_bytes = need; return (scan_fn)k_scan<QB, M, kUDeep, E, kKMax, wps_of(M), 1>; }
    FIR_DEEP(1, 0, kEpiTop1) FIR_DEEP(1, 1, kEpiTop1) FIR_DEEP(1, 2, kEpiTop1)
    FIR_DEEP(1, 0, kEpiStore) FIR_DEEP(1, 1, kEpiStore) FIR_DEEP(1, 2, kEpiStore)
    FIR_DEEP(1, 3, kEpiTop1) FIR_DEEP(1, 4, kEpiTop1) FIR_DEEP(1, 3, kEpiStore) FIR_DEEP(1, 4, kEpiStore)
#undef FIR_DEEP
#endif
    return nullptr;
}

scan_fn pick(int epi, int qb, int metric) {
    switch (epi) {
        case kEpiTop1: return pick_kernel<kEpiTop1>(qb, metric);
        case kEpiTopK: return pick_kernel<kEpiTopK>(qb, metric);
        default: return pick_kernel<kEpiStore>(qb, metric);
    }
}

// Number of waves for `tiles` tiles. All waves are resident at once (waves <= capacity) and each
// gets ceil(tiles / waves) or one fewer tile. Large galleries use a whole number of waves per
// SIMD -- 3 measured best on MI355X for the HBM-bound L2 scan (profiles/r01_sweep_notes.md):
// with a fractional count the SIMDs that hold one wave more finish late.
int pick_waves(int64_t tiles, int max_waves, int simds) {
    if (tiles <= 0) return 4;
    const int per_simd = std::max(1, std::min(3, max_waves / std::max(simds, 1)));
    const int w = per_simd * simds;
    if (tiles <= w) return (int)((tiles + 3) / 4 * 4);
    return w;
}

// Queries per gallery pass. Galleries larger than the 256 MiB Infinity Cache are streamed from HBM and the
// L2 scan stays HBM-bound up to 8 queries per pass (profiles/r01_sweep_notes.md); a gallery (shard) that stays
// cache-resident is VALU-bound either way, and 16 queries per pass halve its cache traffic
// (profiles/r01_qb_table_100kx512.txt: 295k vs 175k queries/s at 100k x 512).
double gallery_bytes(const fir_gallery* g) { return (double)g->tiles * 64.0 * g->dp4 * 16.0; }
// Galleries up to this size are read with plain loads instead of the non-temporal hint. 0: measured no gain from
// plain loads even for an 18 MB gallery that fits the L2s (one-query calls unchanged, multi-pass calls slower --
// profiles/r01_small_gallery_sweep.txt), so every gallery is streamed.
constexpr double kL2ResidentBytes = 0.0;

int effective_qpp(const fir_gallery* g) {
    if (g->qpp > 0) return g->qpp;
    return gallery_bytes(g) <= 384.0 * 1024 * 1024 ? 16 : 8;
}

// Queries per pass of one top-1 call of qb queries. A wave owns whole 64-row tiles, so a gallery of few tiles gives
// few waves per pass: the automatic choice halves the query tile until the launch (tiles x passes) has a wave for
// every SIMD (profiles/r01_small_gallery_sweep.txt).
int top1_qpp(const fir_gallery* g, int qb, int cap) {
    if (g->qpp > 0) return cap;
    const int64_t simds = (int64_t)g->cus * 4;
    int q = cap;
    const int64_t tiles = g->tiles_limit > 0 ? std::min<int64_t>(g->tiles_limit, g->tiles) : g->tiles;
    while (q > 1 && tiles * ((qb + q - 1) / q) < simds) q /= 2;
    return q;
}

// Automatic matrix-core dispatch (fir_gallery_set_large_batch_mfma): measured on MI355X, 1M x 512: 256 queries 444k/s
// against 26k/s through the exact scan, identical keys (profiles/); below ~64k rows the gallery is cache-resident and the
// scan's 16-queries-per-pass form is within reach of it for small batches: a cost model of the two forms decides there.
constexpr int kAutoMfmaQueries = 128;
constexpr int64_t kAutoMfmaRows = 65536;
bool wants_mfma(const fir_gallery* g, int32_t qb, int32_t start, int32_t end) {
    // the whole row, or a prefix of whole 16-feature k-blocks (>= 64 features: below that the scan's prefix pass is cheap)
    if (g->metric != FIR_METRIC_L2 || start != 0 || g->n <= 0 || g->tiles_limit > 0) return false;
    if (end != g->d && (end < 64 || end % 16 != 0)) return false;
    if (g->large_batch_min == 0 || g->gemm_failed) return false;
    if (g->large_batch_min > 0) return qb >= g->large_batch_min;
    if (g->qpp != 0) return false;                                              // a pinned queries-per-pass asks for the scan
    if (g->n < kAutoMfmaRows) {
        // Cache-resident galleries: the scan folds its 16-query passes into one launch, ~40 us + (2 us + 0.25 us per MB) per pass;
        // the matrix-core call ~125 us (225 us with rows longer than 512 features: streamed query slabs) + 0.06 (0.12) us per MB and
        // 128 queries. Measured (us, scan / matrix cores): 40 000 x 512: 128 queries 329 / 140, 1 024: 1 568 / 178; 16 384 x 512:
        // 128: 121 / 135, 256: 234 / 138, 1 024: 788 / 155; 8 192 x 512: 256: 105 / 133, 1 024: 303 / 145; 30 000 x 1536: 128:
        // 589 / 387, 1 024: 4 036 / 496; 3 030 x 1536 (the reference's gallery): 256: 136 / 270, 1 024: 377 / 346. The matrix cores
        // are taken where the model gives them 15 % or more.
        if (g->n < 2048 || qb < 64) return false;
        // (the constants were measured on a 256-CU MI355X; the per-MB terms -- bandwidth and matrix-core time -- scale with the CU count
        // of the device the gallery lives on, the fixed terms -- launches, synchronisation -- do not)
        const double mbs = (double)g->n * (double)end * 4.0 / 1.0e6 * (256.0 / std::max(g->cus, 1));
        const bool streamed = end > 512;
        // (round 4: every super-batch finds its threshold on the way and small calls stay on one stream -- the matrix-core call's fixed part
        // fell from ~125 / 225 us to ~100 / 170 us: 8 192 x 512, 128 / 256 / 1 024 queries 103 / 100 / 115 us, 65 536 x 512 98 / 99 / 162,
        // 8 192 x 1280 148 / 200 / 231, 65 536 x 1280 172 / 234 / 446: profiles/r04_adaptive_cutoff.txt)
        const double mfma_us = (streamed ? 170.0 : 100.0) + (double)((qb + 127) / 128) * (streamed ? 0.12 : 0.06) * mbs;
        const double scan_us = 40.0 + (double)((qb + 15) / 16) * (2.0 + 0.25 * mbs);
        return mfma_us * 1.15 < scan_us;
    }
    // Smaller batches on larger galleries: a matrix-core call costs ~125 us + 0.1 us per MB of compared rows whatever the batch
    // (up to 128 queries), the scan 0.15 us per MB for every 8 queries. Measured at d = 512 (one MI355X, device pointers):
    // 1M rows 8 / 16 / 64 queries: scan 397 / 750 / 2583 us, matrix cores 347 / 336 / 343; 100 000 rows 16 / 32 / 64: 108 / 283 /
    // 445 against 146 / 148 / 159; 65 536 rows 32 / 64: 134 / 429 against 133 / 148; 1M rows 2 / 4 / 7 queries: 349 / 353 / 1016 against ~345.
    const double mb = (double)g->n * (double)end * 4.0 / 1.0e6 * (256.0 / std::max(g->cus, 1));     // (scaled as above)
    if (qb >= kAutoMfmaQueries || (qb >= 32 && mb >= 128.0)) return true;
    if (qb < 2 || mb < 300.0) return false;
    // 2..31 queries over rows streamed from HBM: the scan takes one pass per power-of-two group of up to 8 queries (3 queries: 2 + 1,
    // 7: 4 + 2 + 1), ~15 us + 0.16 us per MB each (1M x 512, 2 048 MB: 340 us for 1, 2, 4 or 8 queries, three times that for 7); the
    // matrix-core call, since round 4 (live query blocks only, one stream, the first row block summed once: profiles/r04_small_calls.txt),
    // ~75 us + 0.095 us per MB for anything up to 32 queries (1M x 512: 265 us; 1M x 1280: 575 us), ~125 us + 0.105 us per MB up to 128
    const int passes = qb / 8 + __builtin_popcount((unsigned)qb & 7u);
    return (qb <= 32 ? 75.0 + 0.095 * mb : 125.0 + 0.105 * mb) < passes * (15.0 + 0.16 * mb);
}

// ONE query against rows far beyond the caches (automatic mode only): the nomination scan over the fp16 copy
// (fir_gemm_search_few_keys_dev) reads half the bytes of the exact scan's pass and costs ~65 us more around it: 1M x 512 (2 GB)
// 332 -> 253 us, 1M x 1280 790 -> 550 us, 300 000 x 512 116 -> 173 us (not taken). With 2 queries the two forms tie, from 3 on
// the dot products (v_dot2 + one LDS read per query and fragment) cost more than the bytes saved: those stay with the f32 scan.
bool wants_few(const fir_gallery* g, int32_t qb, int32_t start, int32_t end) {
    if (g->metric != FIR_METRIC_L2 || start != 0 || g->n < kAutoMfmaRows || g->tiles_limit > 0 || g->qpp != 0 || g->few_hits < 0) return false;
    if (end != g->d && (end < 64 || end % 16 != 0)) return false;
    if (g->large_batch_min >= 0 || g->gemm_failed) return false;
    return qb == 1 && (double)g->n * (double)end * 4.0 >= 1500.0e6;
}

void note_dispatch(fir_gallery* g, const void* fn, const char* name, int launches_add, int gx, int gy, int block, size_t dyn_lds, int qpp,
                   double bytes, double flops, int path) {
    fir_dispatch_info& L = g->last;
    if (launches_add == 0 || std::strncmp(L.kernel, name, sizeof(L.kernel)) != 0) {
        std::memset(&L, 0, sizeof L);
        std::snprintf(L.kernel, sizeof(L.kernel), "%s", name);
        hipFuncAttributes at;
        if (fn && hipFuncGetAttributes(&at, fn) == hipSuccess) {
            L.vgprs = at.numRegs;
            L.lds_bytes = (int32_t)(at.sharedSizeBytes + dyn_lds);
        } else {
            L.lds_bytes = (int32_t)dyn_lds;
        }
    }
    L.struct_bytes = (int32_t)sizeof L;
    L.path = path;
    L.launches += 1;
    L.grid_x = gx; L.grid_y = gy; L.block = block;
    L.queries_per_pass = qpp;
    L.bytes_per_launch = bytes;
    L.flops_per_launch = flops;
}

int check_range(const fir_gallery* g, int32_t& start, int32_t& end) {
    if (end == 0) end = g->d;   // db_features.cpp:320-321
    if (start < 0 || end > g->d || start >= end)
        return fail(FIR_ERR_ARG, "feature range [%d,%d) not inside [0,%d)", start, end, g->d);
    return FIR_OK;
}

// Serial number of a query transposition (fir_gallery::range): wraps without overflow; a collision after 2^32 calls only
// sends one chi-square / KL call through the full division sequence.
int next_serial(fir_gallery* g) {
    g->q_serial = (int)((unsigned)g->q_serial + 1u);
    return g->q_serial;
}

int largest_pow2_le(int x, int cap) {
    int p = 1;
    while (p * 2 <= x && p * 2 <= cap) p *= 2;
    return p;
}

// Resident-wave capacity of one scan kernel on this device (cached per kernel / LDS size).
int max_waves_for(fir_gallery* g, scan_fn fn, size_t lds_bytes) {
    for (const auto& e : g->occ)
        if (e.fn == (const void*)fn && e.lds == lds_bytes) return e.waves;
    int nb = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)fn, kBlock, lds_bytes);
    if (e != hipSuccess || nb <= 0) nb = 2;
    nb = std::min(nb, 8);
    const int waves = g->cus * nb * (kBlock / 64);
    g->occ.push_back({(const void*)fn, lds_bytes, waves});
    return waves;
}

// Queue: transpose (+ key init) of one query tile, then one gallery pass.
int run_pass(fir_gallery* g, hipStream_t st, int epi, const float* d_queries, int q0, int qb_tile, int32_t start,
             int32_t end, uint64_t* keys, float* out, int64_t out_stride, int k, int* waves_used = nullptr, int ny = 1,
             int init_keys = 0) {
    // ny > 1 (hand-scheduled top-1 kernels only): ny consecutive query tiles of qb_tile queries in ONE launch
    const int kk = g->dp4 * 4;
    float* qt = g->qt + (size_t)q0 * kk;
    {
        const int64_t total = std::max<int64_t>((int64_t)kk * qb_tile * ny, init_keys);
        const int blocks = (int)((total + kBlock - 1) / kBlock);
        hipLaunchKernelGGL(k_transpose_queries, dim3(blocks), dim3(kBlock), 0, st, d_queries + (size_t)q0 * g->d, qb_tile * ny,
                           g->d, g->dp4, qb_tile, qt, init_keys > 0 ? keys : nullptr, init_keys, g->range, next_serial(g));
    }
    size_t lds_bytes = 0;
    char kname[160];
    scan_fn fn = pick_fast(epi, qb_tile, g->metric, start, end, g->dp4, &lds_bytes);
    if (fn) std::snprintf(kname, sizeof kname, "fir::%s<%d, %d, %d, false>", lds_bytes ? "k_scan_l2_lds" : "k_scan_l2_fast", qb_tile / 8, FIR_FAST_U, FIR_FAST_WPS);
    if (!fn && g->tiles <= (int64_t)g->cus * 4) {
        fn = pick_deep(epi, qb_tile, g->metric, g->dp4, &lds_bytes);
        if (fn) std::snprintf(kname, sizeof kname, "fir::k_scan<%d, %d, %d, %d, %d, %d, 1>", qb_tile, g->metric, kUDeep, epi, kKMax, wps_of(g->metric));
    }
    if (!fn) {
        fn = pick(epi, qb_tile, g->metric);
        std::snprintf(kname, sizeof kname, "fir::k_scan<%d, %d, %d, %d, %d, %d, 0>", qb_tile, g->metric, kU, epi, kKMax, wps_of(g->metric));
    }
    if (!fn) return fail(FIR_ERR_ARG, "no kernel for qb=%d metric=%d", qb_tile, g->metric);
    if (lds_bytes > 64 * 1024) {   // more than the default dynamic LDS limit: opt in once per kernel
        bool known = false;
        for (const auto& e : g->occ) known = known || (e.fn == (const void*)fn && e.lds == lds_bytes);
        if (!known) FIR_HIP(hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxQueryTileLds));
    }
    // chi-square / KL: the plain-range twin of the kernel, launched next to it with the same grid -- which of the two
    // does the pass is decided on the device from the range flags (k_scan), the other returns at once
    scan_fn fn_plain = nullptr;
    if (g->metric != kL2) {
        size_t lds2 = 0;
        if (g->tiles <= (int64_t)g->cus * 4 && lds_bytes > 0) fn_plain = pick_deep(epi, qb_tile, g->metric + 2, g->dp4, &lds2);
        if (!fn_plain && lds_bytes == 0) fn_plain = pick(epi, qb_tile, g->metric + 2);
    }
    int max_waves = max_waves_for(g, fn, lds_bytes);
    if (fn_plain) max_waves = std::min(max_waves, max_waves_for(g, fn_plain, lds_bytes));
    int waves = g->waves_req > 0 ? std::min(g->waves_req, max_waves) : pick_waves(g->tiles, max_waves, g->cus * 4);
    const int groups = epi == kEpiTop1 && g->sample_groups > 1 && g->metric != kL2 ? g->sample_groups : 0;
    if (groups) waves = std::max(4 * groups, waves / (4 * groups) * (4 * groups));      // a wave's tiles all belong to one group
    g->last_waves = waves;
    if (waves_used) *waves_used = waves;
    ScanArgs a{};
    a.groups = groups;
    a.group_stride = g->sample_group_stride;
    a.qt = qt;
    const int64_t tile0 = g->tiles_limit > 0 ? std::min<int64_t>(g->tile_begin, g->tiles) : 0;
    const int64_t tiles = g->tiles_limit > 0 ? std::min<int64_t>(g->tiles_limit, g->tiles - tile0) : g->tiles;
    a.gal4 = g->gal4 + (size_t)tile0 * g->dp4 * 64;
    a.row_offset = g->row_offset + tile0 * kTileRows;
    a.n = std::min<int64_t>(g->n - tile0 * kTileRows, tiles * kTileRows);
    a.tiles = (int32_t)tiles;
    a.dp4 = g->dp4;
    a.start = start;
    a.end = end;
    a.waves = waves;
    a.keys = keys;
    a.out = out;
    a.out_stride = out_stride;
    a.nq = qb_tile;
    a.k = k;
    a.qt_stride = (int64_t)kk * qb_tile;
    a.nt = gallery_bytes(g) > kL2ResidentBytes ? 1 : 0;
    a.range = g->range;
    a.serial = g->q_serial;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (g->profiling) {
        if (g->ev_used + 2 > g->ev.size()) {
            for (int i = 0; i < 64; ++i) {
                hipEvent_t e;
                FIR_HIP(hipEventCreate(&e));
                g->ev.push_back(e);
            }
        }
        e0 = g->ev[g->ev_used++];
        e1 = g->ev[g->ev_used++];
        FIR_HIP(hipEventRecord(e0, st));
    }
    hipLaunchKernelGGL(fn, dim3(waves / 4, ny), dim3(kBlock), lds_bytes, st, a);
    if (fn_plain) hipLaunchKernelGGL(fn_plain, dim3(waves / 4, ny), dim3(kBlock), lds_bytes, st, a);
    // algorithmic bytes of one launch: per pass the gallery range once, the query tile, the keys
    const double bytes_alg = ny * ((double)a.n * (end - start) * 4.0 + (double)qb_tile * (end - start) * 4.0 + qb_tile * 8.0);
    if (g->profiling) {
        FIR_HIP(hipEventRecord(e1, st));
        g->last_bytes = bytes_alg;
    }
    if (!g->quiet) note_dispatch(g, (const void*)fn, kname, g->call_launches++, waves / 4, ny, kBlock, lds_bytes, qb_tile, bytes_alg, 0.0, 0);
    FIR_HIP(hipGetLastError());
    return FIR_OK;
}

int topk_lists_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys,
                   hipStream_t st);

int top1_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, uint64_t* d_keys,
             hipStream_t st) {
    // chi-square batches over a large plain-range gallery: nomination scan + exact re-rank (topk_lists_dev with K = 1), the same
    // keys at about twice the rate; it synchronises `st` once (it has to know that no list overflowed) and leaves the call to the
    // exact scan below when it cannot answer (operands outside the plain range, list overflow)
    if ((g->metric == kChi2 || g->metric == kKL) && g->gallery_plain && g->tiles_limit == 0 && !g->quiet && qb >= 8 && g->n >= 65536 &&
        g->qpp == 0) {
        int rc0 = FIR_OK;
        for (int q0 = 0; q0 < qb && rc0 == FIR_OK; q0 += 1024)       // candidate lists are 32 KiB per query: bounded scratch for any qb
            rc0 = topk_lists_dev(g, d_queries + (size_t)q0 * g->d, std::min(1024, qb - q0), start, end, 1, d_keys + q0, st);
        if (rc0 != FIR_ERR_STATE) return rc0;
    }
    int rc = grow(g->qt, g->qt_cap, (size_t)qb * g->dp4 * 4 + 64);   // +64: the fast kernel prefetches one unit past the tile
    if (rc) return rc;
    // 16 queries per pass exist only in the hand-scheduled kernel (whole-chunk L2 ranges)
    int cap = pick_fast(kEpiTop1, 16, g->metric, start, end, g->dp4, nullptr) ? effective_qpp(g) : std::min(effective_qpp(g), 8);
    // 16 queries per pass only while their tile fits the default 64 KiB of LDS (d <= 1020): a larger tile leaves one
    // workgroup per CU, and the 8-query tile then does better
    if (cap > 8 && (size_t)(g->dp4 + 1) * 16 * 16 > 64 * 1024) cap = 8;
    cap = top1_qpp(g, qb, cap);
    int q0 = 0;
    while (q0 < qb) {
        const int t = largest_pow2_le(qb - q0, cap);
        // all the whole tiles of t queries go into one launch (blockIdx.y)
        const int ny = std::min((qb - q0) / t, g->max_tiles_per_launch);
        rc = run_pass(g, st, kEpiTop1, d_queries, q0, t, start, end, d_keys + q0, nullptr, 0, 0, nullptr, ny, /*init_keys=*/t * ny);
        if (rc) return rc;
        q0 += t * ny;
    }
    return FIR_OK;
}

int topk_lists_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys,
                   hipStream_t st);

int try_mfma_topk(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys, hipStream_t st);

int topk_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys,
             hipStream_t st, bool allow_lists = true, bool allow_mfma = true) {
    // large whole-range L2 batches over a large gallery: the matrix-core nomination pass + exact re-rank (fir_gemm.hip), the
    // same keys; what it cannot certify comes back through this function with allow_mfma = false
    if (allow_mfma) {
        const int rcm = try_mfma_topk(g, d_queries, qb, start, end, k, d_keys, st);
        if (rcm <= 0) return rcm;
    }
    // batches over a large gallery: threshold from a row sample, append scan at the speed of the top-1 scan, K smallest
    // of each candidate list (exact distances throughout); anything it cannot certify falls back to the scan below
    if (allow_lists && g->tiles_limit == 0 && qb >= 8 && g->n >= 65536) {
        constexpr int kListBatch = 1024;           // candidate lists are 32 KiB per query: bounded scratch for any qb
        if (qb > kListBatch) {
            for (int q0 = 0; q0 < qb; q0 += kListBatch) {
                const int rc3 = topk_dev(g, d_queries + (size_t)q0 * g->d, std::min(kListBatch, qb - q0), start, end, k, d_keys + (size_t)q0 * k, st, true, false);
                if (rc3) return rc3;
            }
            return FIR_OK;
        }
        const int rc2 = topk_lists_dev(g, d_queries, qb, start, end, k, d_keys, st);
        if (rc2 != FIR_ERR_STATE) return rc2;      // FIR_ERR_STATE: not certified -> the register-list scan answers
    }
    int rc = grow(g->qt, g->qt_cap, (size_t)qb * g->dp4 * 4);
    if (rc) return rc;
    const int qcap = std::min(effective_qpp(g), 4);   // 2*kKMax registers per query per lane
    rc = grow(g->part, g->part_cap, (size_t)g->max_waves * qcap * k);
    if (rc) return rc;
    int q0 = 0;
    while (q0 < qb) {
        const int t = largest_pow2_le(qb - q0, qcap);
        int waves = 0;
        rc = run_pass(g, st, kEpiTopK, d_queries, q0, t, start, end, g->part, nullptr, 0, k, &waves);
        if (rc) return rc;
        hipLaunchKernelGGL(k_topk_merge, dim3(t), dim3(kBlock), 0, st, g->part, waves, t, q0, t, k, d_keys);
        q0 += t;
    }
    FIR_HIP(hipGetLastError());
    return FIR_OK;
}

// The K nearest rows of a batch through candidate lists (see k_scan_l2_lds<..., APPEND>). Returns FIR_ERR_STATE (without
// setting the error text) when a query could not be certified: fewer than K sample rows below 100000, or a list overflow.
constexpr int kListCap = 4096;
int topk_lists_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys,
                   hipStream_t st) {
    // the hand-scheduled LDS-tile kernel where it applies (L2, whole-chunk ranges, tile within 64 KiB), else the generic one
    size_t lds_bytes = 0;
    scan_fn probe = pick_fast(kEpiTop1, 8, g->metric, start, end, g->dp4, &lds_bytes);
    const bool fast = probe && lds_bytes > 0 && lds_bytes <= 64 * 1024;
    if (!fast) lds_bytes = 0;
    // chi-square over a plain-range gallery: the append scan runs a NOMINATION metric (1-ulp reciprocal, 5.5 instead of 11 issue
    // slots per element) against a threshold widened by its error bound, and the few hundred appended rows per query are
    // re-ranked with the reference's arithmetic before the K smallest are taken: the same keys as the exact scan, about twice
    // as fast. Every row whose reference distance is <= the unwidened threshold is appended (approx <= exact (1 + eps)), and
    // the K-th smallest reference distance is <= that threshold (K sample rows are), so the K best are all in the list.
    const bool no_nominate = fir_knob_("FIR_NO_CHI2_NOMINATION") != nullptr;      // experiments
    const bool nominate = g->metric == kChi2 && g->gallery_plain && !no_nominate && (size_t)g->d * sizeof(float) <= 48 * 1024;
    // KL over a plain-range gallery, the entropy form: KL = ln2 (Lq + Lg - E), Lq = sum(l log2 l + l), Lg the same over the row (both
    // added up in double once per query / per row), E = sum_k s_k log2 s_k with s_k = l_k + r_k the only sum the scan runs: a packed
    // add, a v_log_f32 and a packed fma per (value, query) = 3 issue slots instead of the ~35 of two quotients and two logarithms.
    // Error against the reference's float value, u = 2^-24, S = sum(l) + sum(r), |log2 s_k| <= 26 in the plain range:
    //   rounding of s_k: (26 + log2 e) u s_k;  v_log_f32, 1 ulp of a value below 32: 32 u s_k;  the fmas of a 4 U-term group and
    //   the tail / edge terms: <= 64 * 26.1 u S;  the nf / (4 U) group sums: (nf / 32) 26.1 u S;  Lq, Lg to float and the two
    //   combining operations: <= 85 u S;  the reference's own chain (two products and two adds per feature, |terms| <= S): (2 nf + 8) u S
    //   |entropy form - reference| <= B = [ln2 (26.1 (64 + nf / 32) + 150) + 2 nf + 8] 2^-24 (sum(l) + max_rows sum(r)) / nf.
    // The threshold is widened by 1.5 B (k_query_entropy_widen), the appended rows are re-ranked with the exact scan's arithmetic.
    const bool klent = g->metric == kKL && g->gallery_plain && !no_nominate && (size_t)g->d * sizeof(float) <= 48 * 1024 && FIR_U == 8;
    // Two nomination metrics. kChi2Approx: (l - r)^2 * rcp(l + r), within (2 nf + 8) 2^-24 RELATIVE of the reference's value (all
    // terms >= 0), 5.5-6.6 issue slots per element. kChi2Harm (default): chi2 = sum(l) + sum(r) - 4 sum_k 1/(1/l_k + 1/r_k), two
    // terms per reciprocal, 1/A + 1/B = (A + B) / (A B) with A = 1/l_k + 1/r_k: 2.25 issue slots per element, 1/r computed once per
    // gallery value and pass. Its error is relative to sum(l) + sum(r), not to chi2, u = 2^-24: 1/l (IEEE division) u, 1/r (v_rcp_f32,
    // 1 ulp) 2 u, so A and B within 3 u, A + B 4 u, A B 7 u, its reciprocal 9 u, the pair of terms (4 + 9 + 1) u = 14 u of its value;
    // harmonic terms <= (l + r)/4, so 4 * their sum <= sum(l) + sum(r): 14 u (sum(l) + sum(r)); the fma chain of nf / 2 non-negative
    // terms adds nf u / 2, the two plain sums nf u each, the reference's own chain (nf + 3) u of chi2 <= sum(l) + sum(r):
    //   |harmonic form - reference| <= B = (3 nf + 19) 2^-24 (sum(l) + max_rows sum(r)) / nf.
    // The threshold is widened by 1.5 B (k_query_sums_widen): every row whose reference distance is within the unwidened one is appended.
    const char* form_env = fir_knob_("FIR_CHI2_NOMINATION");                                   // experiments / tests: 1 = kChi2Approx
    const int chi2_form = form_env ? std::atoi(form_env) : 2;
    const bool harm = nominate && chi2_form == 2;
    const float tau_scale = nominate && !harm ? 1.0f + 1.5f * (2.0f * (float)(end - start) + 16.0f) * 5.9604645e-8f : 1.0f;
    // the two cheap nomination forms take two tiles of 8 queries per gallery read (k_nominate): whole pairs of tiles
    const int nh = (harm || klent) && !fir_knob_("FIR_NOMINATE_ONE_TILE") ? 2 : 1;
    const int qpad = (qb + 8 * nh - 1) / (8 * nh) * (8 * nh);
    void *p_skeys = nullptr, *p_small = nullptr, *p_lists = nullptr;
    int rc;
    if ((rc = fir_gallery_scratch_(g, 12, (size_t)qpad * k * 8, &p_skeys))) return rc;
    if ((rc = fir_gallery_scratch_(g, 13, (size_t)qpad * 8 + 16, &p_small))) return rc;
    if ((rc = fir_gallery_scratch_(g, 14, (size_t)qpad * kListCap * 8, &p_lists))) return rc;
    uint64_t* skeys = (uint64_t*)p_skeys;
    float* tau = (float*)p_small;
    int32_t* counts = (int32_t*)(tau + qpad);
    int32_t* flag = counts + qpad;
    uint64_t* lists = (uint64_t*)p_lists;
    if ((rc = grow(g->qt, g->qt_cap, (size_t)qpad * g->dp4 * 4 + 64))) return rc;      // before anything is queued on it
    float* sq = nullptr;
    if (harm || klent) {
        void* p_sq = nullptr;
        if ((rc = fir_gallery_scratch_(g, 15, (size_t)qpad * sizeof(float), &p_sq))) return rc;
        sq = (float*)p_sq;
        if (g->rs_start != start || g->rs_end != end || g->rs_metric != g->metric || !g->rowsum) {
            if ((rc = grow(g->rowsum, g->rowsum_cap, (size_t)g->n + 4))) return rc;
            FIR_HIP(hipMemsetAsync(g->rowsum + g->n, 0, 4 * sizeof(float), st));
            hipLaunchKernelGGL(klent ? k_row_entropy : k_row_sums, dim3((unsigned)((g->n + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, g->gal4, g->n, g->dp4,
                               start, end, g->rowsum, (unsigned int*)(g->rowsum + g->n));
            g->rs_start = start;
            g->rs_end = end;
            g->rs_metric = g->metric;
        }
    }
    // 1. nearest row inside each of k disjoint groups of sample tiles (k top-1 scans, each over all the queries): the
    //    largest of the k distances is a threshold at least k rows pass; about 2.3 * n * k / rows_sampled rows will
    const int64_t want_rows = std::max<int64_t>(16384, (int64_t)g->n * k / 256);
    const int64_t group_tiles = std::max<int64_t>(1, std::min<int64_t>(g->tiles / k, (want_rows / k + kTileRows - 1) / kTileRows));
    // (the row samples are not what a profile of this call is about: the events and the dispatch record belong to the append scan below)
    const bool was_profiling = g->profiling, was_quiet = g->quiet;
    g->profiling = false;
    g->quiet = true;
    FIR_HIP(hipMemsetAsync(flag, 0, 4, st));
    // chi-square / KL nomination: the row samples run the NOMINATION metric too (2.25 / 3 issue slots per element instead of the
    // exact metric's 11 / ~35: the samples were 0.9 of a 256-query call's 12.3 ms). A sampled minimum is then within B of the
    // reference's value of that row, so at least K rows have a reference distance <= max_i(sample_i) + B, and the append scan
    // (the same metric) has to take everything <= max_i(sample_i) + 2 B: the thresholds are widened by 2.5 B instead of 1.5 B.
    const bool approx_samples = (harm || klent) && !fir_knob_("FIR_EXACT_SAMPLES");
    const int kk_q = g->dp4 * 4;
    const int sample_stride = approx_samples ? qpad : qb;                  // keys of sample group i: skeys[i * sample_stride + q]
    bool tiles_ready = false;                                              // the query tiles are already in g->qt, in the nomination form
    if (approx_samples) {
        const float nf = (float)(end - start);
        const float coef = harm ? (3.0f * nf + 19.0f) * 5.9604645e-8f / nf
                                : (0.6932f * (26.1f * (64.0f + nf / 32.0f) + 150.0f) + 2.0f * nf + 8.0f) * 5.9604645e-8f / nf;
        // the per-query sums first (thresholds at -inf: nothing to widen yet)
        FIR_HIP(hipMemsetD32Async((hipDeviceptr_t)tau, (int)0xFF800000u, (size_t)qpad, st));
        hipLaunchKernelGGL(harm ? k_query_sums_widen : k_query_entropy_widen, dim3(qpad), dim3(64), 0, st, d_queries, qb, g->d, start, end, sq, tau,
                           (const unsigned int*)(g->rowsum + g->n), coef);
        // every query tile of the call, once, in the form both scans read
        hipLaunchKernelGGL(k_transpose_queries, dim3((unsigned)(((int64_t)kk_q * qpad + kBlock - 1) / kBlock)), dim3(kBlock), 0, st, d_queries, qb, g->d,
                           g->dp4, 8, g->qt, (uint64_t*)nullptr, 0, g->range, next_serial(g), harm ? 1 : 2);
        tiles_ready = true;
        FIR_HIP(hipMemsetAsync(skeys, 0xFF, (size_t)qpad * k * 8, st));
        const scan_fn sfn = harm ? (scan_fn)k_scan<8, kChi2Harm, kU, kEpiTop1, kKMax, kWpsPlain> : (scan_fn)k_scan<8, kKLEnt, kU, kEpiTop1, kKMax, kWpsPlain>;
        const int64_t tiles = std::min<int64_t>(group_tiles * k, g->tiles);
        int waves = pick_waves(tiles, max_waves_for(g, sfn, 0), g->cus * 4);
        if (k > 1) waves = std::max(4 * k, waves / (4 * k) * (4 * k));                  // a wave's tiles all belong to one group
        ScanArgs a{};
        a.groups = k > 1 ? k : 0;
        a.group_stride = sample_stride;
        a.qt = g->qt;
        a.gal4 = g->gal4;
        a.row_offset = g->row_offset;
        a.n = std::min<int64_t>(g->n, tiles * kTileRows);
        a.tiles = (int32_t)tiles;
        a.dp4 = g->dp4;
        a.start = start;
        a.end = end;
        a.waves = waves;
        a.keys = skeys;
        a.nq = 8;
        a.qt_stride = (int64_t)kk_q * 8;
        a.nt = 0;
        a.range = g->range;
        a.serial = g->q_serial;
        a.flag = flag;
        a.sg = g->rowsum;
        a.sq = sq;
        for (int y0 = 0; y0 < qpad / 8; y0 += g->max_tiles_per_launch) {
            const int ny = std::min(g->max_tiles_per_launch, qpad / 8 - y0);
            ScanArgs b = a;
            b.qt = g->qt + (size_t)y0 * 8 * kk_q;
            b.keys = skeys + (size_t)y0 * 8;
            b.sq = sq + (size_t)y0 * 8;
            hipLaunchKernelGGL(sfn, dim3(waves / 4, ny), dim3(kBlock), 0, st, b);
        }
    } else if (k > 1 && g->metric != kL2) {
        // chi-square / KL: the K samples in ONE launch -- sample tile t belongs to group t mod K, a wave reports into its group's keys
        // (K launches of group_tiles tiles each leave most of the chip idle: a tile is one wave's serial work)
        FIR_HIP(hipMemsetAsync(skeys, 0xFF, (size_t)qb * k * 8, st));
        g->tile_begin = 0;
        g->tiles_limit = group_tiles * k;
        g->sample_groups = k;
        g->sample_group_stride = qb;
        rc = top1_dev(g, d_queries, qb, start, end, skeys, st);
        g->sample_groups = 0;
    } else {
        for (int i = 0; i < k && !rc; ++i) {
            g->tile_begin = i * group_tiles;
            g->tiles_limit = group_tiles;
            rc = top1_dev(g, d_queries, qb, start, end, skeys + (size_t)i * qb, st);
        }
    }
    g->tiles_limit = 0;
    g->tile_begin = 0;
    g->profiling = was_profiling;
    g->quiet = was_quiet;
    if (rc) return rc;
    hipLaunchKernelGGL(k_topk_tau, dim3((qpad + 63) / 64), dim3(64), 0, st, skeys, qb, qpad, k, tau, counts, flag, tau_scale, sample_stride);
    // (1.5 B over an exact sample; 2.5 B over a sample in the nomination metric -- the kernels multiply by 1.5)
    const float widen = approx_samples ? 1.67f : 1.0f;
    if (harm) {
        const float nf = (float)(end - start);
        hipLaunchKernelGGL(k_query_sums_widen, dim3(qpad), dim3(64), 0, st, d_queries, qb, g->d, start, end, sq, tau, (const unsigned int*)(g->rowsum + g->n),
                           widen * (3.0f * nf + 19.0f) * 5.9604645e-8f / nf);
    }
    if (klent) {
        const float nf = (float)(end - start);
        hipLaunchKernelGGL(k_query_entropy_widen, dim3(qpad), dim3(64), 0, st, d_queries, qb, g->d, start, end, sq, tau, (const unsigned int*)(g->rowsum + g->n),
                           widen * (0.6932f * (26.1f * (64.0f + nf / 32.0f) + 150.0f) + 2.0f * nf + 8.0f) * 5.9604645e-8f / nf);
    }
    // 2. the append scan over the whole gallery: 8 queries per tile, every tile of the call in one launch (blockIdx.y)
    const int kk = g->dp4 * 4;
    scan_fn fn = fast ? (scan_fn)k_scan_l2_lds<1, FIR_FAST_U, FIR_FAST_WPS, true>
                      : klent ? (scan_fn)k_scan<8, kKLEnt, kU, kEpiAppend, kKMax, kWpsPlain>
                      : harm ? (scan_fn)k_scan<8, kChi2Harm, kU, kEpiAppend, kKMax, kWpsPlain>
                      : nominate ? (scan_fn)k_scan<8, kChi2Approx, kU, kEpiAppend, kKMax, kWpsPlain>
                      : g->metric == kL2 ? (scan_fn)k_scan<8, kL2, kU, kEpiAppend, kKMax, kWps>
                      : g->metric == kChi2 ? (scan_fn)k_scan<8, kChi2, kU, kEpiAppend, kKMax, kWps>
                                           : (scan_fn)k_scan<8, kKL, kU, kEpiAppend, kKMax, kWps>;
    const scan_fn fn_plain = nominate || klent ? nullptr
                           : g->metric == kChi2 ? (scan_fn)k_scan<8, kChi2InRange, kU, kEpiAppend, kKMax, kWpsPlain>
                           : g->metric == kKL ? (scan_fn)k_scan<8, kKLInRange, kU, kEpiAppend, kKMax, kWpsPlain> : nullptr;   // see run_pass
    if (nh == 2) fn = klent ? (scan_fn)k_nominate<kKLEnt, 2, kU, kWpsPlain> : (scan_fn)k_nominate<kChi2Harm, 2, kU, kWpsPlain>;
    char fn_name[96];
    if (nh == 2) std::snprintf(fn_name, sizeof fn_name, "fir::k_nominate<%d, 2, %d, %d>", klent ? (int)kKLEnt : (int)kChi2Harm, kU, kWpsPlain);
    else if (fast) std::snprintf(fn_name, sizeof fn_name, "fir::k_scan_l2_lds<1, %d, %d, true>", FIR_FAST_U, FIR_FAST_WPS);
    else std::snprintf(fn_name, sizeof fn_name, "fir::k_scan<8, %d, %d, %d, %d, %d>", klent ? (int)kKLEnt : harm ? (int)kChi2Harm : nominate ? (int)kChi2Approx : g->metric, kU,
                       (int)kEpiAppend, kKMax, (nominate || klent) ? kWpsPlain : kWps);
    int launches_noted = 0;
    int max_waves = max_waves_for(g, fn, lds_bytes);
    if (fn_plain) max_waves = std::min(max_waves, max_waves_for(g, fn_plain, lds_bytes));
    const int waves = g->waves_req > 0 ? std::min(g->waves_req, max_waves) : pick_waves(g->tiles, max_waves, g->cus * 4);
    const int tiles_per_launch = nh == 2 ? std::max(2, g->max_tiles_per_launch & ~1) : g->max_tiles_per_launch;
    for (int q0 = 0; q0 < qpad; q0 += 8 * tiles_per_launch) {
        const int ny = std::min(tiles_per_launch, (qpad - q0) / 8);
        const int live = std::max(0, std::min(qb - q0, ny * 8));
        float* qt = g->qt + (size_t)q0 * kk;
        if (!tiles_ready)
            hipLaunchKernelGGL(k_transpose_queries, dim3((unsigned)(((int64_t)kk * 8 * ny + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                               d_queries + (size_t)q0 * g->d, live, g->d, g->dp4, 8, qt, (uint64_t*)nullptr, 0, g->range, next_serial(g), harm ? 1 : klent ? 2 : 0);
        ScanArgs a{};
        a.range = g->range;
        a.serial = g->q_serial;
        a.gal4 = g->gal4;
        a.qt = qt;
        a.n = g->n;
        a.tiles = (int32_t)g->tiles;
        a.dp4 = g->dp4;
        a.start = start;
        a.end = end;
        a.waves = waves;
        a.row_offset = g->row_offset;
        a.keys = lists + (size_t)q0 * kListCap;
        a.k = kListCap;
        a.tau = tau + q0;
        a.counts = counts + q0;
        a.qt_stride = (int64_t)kk * 8;
        a.nt = gallery_bytes(g) > kL2ResidentBytes ? 1 : 0;
        a.flag = flag;
        a.sg = g->rowsum;
        a.sq = harm || klent ? sq + q0 : nullptr;
        // algorithmic bytes of the launch: one read of the compared features per nh tiles of eight queries, the query tiles, the appended keys aside
        const double launch_bytes = (double)(ny / nh) * ((double)g->n * (end - start) * 4.0) + (double)ny * ((double)(end - start) * 32.0 + 64.0);
        if (!g->quiet && (rc = fir_gallery_profile_begin_(g, st))) return rc;
        hipLaunchKernelGGL(fn, dim3(waves / 4, ny / nh), dim3(kBlock), lds_bytes, st, a);
        if (fn_plain) hipLaunchKernelGGL(fn_plain, dim3(waves / 4, ny), dim3(kBlock), lds_bytes, st, a);
        if (!g->quiet) {
            if ((rc = fir_gallery_profile_end_(g, st, launch_bytes))) return rc;
            note_dispatch(g, (const void*)fn, fn_name, launches_noted++, waves / 4, ny / nh, kBlock, lds_bytes, 8 * nh, launch_bytes, 0.0, 0);
        }
    }
    // 3. (nomination) the reference's distance of every appended row, keys rewritten in place
    if (klent)
        hipLaunchKernelGGL(k_list_rerank<kKLInRange>, dim3(qb), dim3(kBlock), (size_t)g->d * sizeof(float), st, lists, counts, kListCap, g->gal4, g->dp4, g->n,
                           g->row_offset, d_queries, g->d, start, end);
    if (nominate)
        hipLaunchKernelGGL(k_list_rerank<kChi2>, dim3(qb), dim3(kBlock), (size_t)g->d * sizeof(float), st, lists, counts, kListCap, g->gal4, g->dp4, g->n,
                           g->row_offset, d_queries, g->d, start, end);
    // 4. the K smallest keys of every list
    hipLaunchKernelGGL(k_topk_select, dim3(qb), dim3(kBlock), 0, st, lists, counts, kListCap, k, d_keys, flag);
    FIR_HIP(hipGetLastError());
    int32_t h_flag = 0;
    FIR_HIP(hipMemcpyAsync(&h_flag, flag, 4, hipMemcpyDeviceToHost, st));
    FIR_HIP(hipStreamSynchronize(st));
    return h_flag ? FIR_ERR_STATE : FIR_OK;
}

int range_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, float* d_out, hipStream_t st) {
    int rc = grow(g->qt, g->qt_cap, (size_t)qb * g->dp4 * 4);
    if (rc) return rc;
    int q0 = 0;
    while (q0 < qb) {
        const int t = largest_pow2_le(qb - q0, std::min(effective_qpp(g), 8));
        rc = run_pass(g, st, kEpiStore, d_queries, q0, t, start, end, nullptr, d_out + (size_t)q0 * g->n, g->n, 0);
        if (rc) return rc;
        q0 += t;
    }
    return FIR_OK;
}

constexpr int kUSub = 8;      // chunks per load group of k_scan_subranges: sub-ranges are multiples of 32 features
scan_fn pick_subranges(int qb, int metric) {
#ifndef FIR_MINIMAL
#define FIR_SUB(QB, M) if (qb == QB && metric == M) return (scan_fn)k_scan_subranges<QB, M, kUSub, kWps>;
    FIR_SUB(1, 0) FIR_SUB(2, 0) FIR_SUB(4, 0) FIR_SUB(8, 0)
    FIR_SUB(1, 1) FIR_SUB(2, 1) FIR_SUB(4, 1) FIR_SUB(8, 1)
    FIR_SUB(1, 2) FIR_SUB(2, 2) FIR_SUB(4, 2) FIR_SUB(8, 2)
#undef FIR_SUB
#endif
    return nullptr;
}

int subranges_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t step, float* d_out, hipStream_t st,
                  int32_t step2 = 0) {
    // step2 != 0: two sub-ranges, [start, start + step) and [start + step, end) with end - start - step == step2 (whole load groups both)
    const int nsub = step2 ? 2 : (end - start) / step;
    if (!step2 && (step % (4 * kUSub) != 0 || start % 4 != 0)) {   // one pass per sub-range (any step)
        for (int ci = 0; ci < nsub; ++ci) {
            const int rc = range_dev(g, d_queries, qb, start + ci * step, start + (ci + 1) * step, d_out + (size_t)ci * qb * g->n, st);
            if (rc) return rc;
        }
        return FIR_OK;
    }
    int rc = grow(g->qt, g->qt_cap, (size_t)8 * g->dp4 * 4);
    if (rc) return rc;
    const int kk = g->dp4 * 4;
    for (int q0 = 0; q0 < qb; q0 += 8) {
        const int live = std::min(8, qb - q0);
        const int qbt = live <= 1 ? 1 : live <= 2 ? 2 : live <= 4 ? 4 : 8;   // kernel tile: the next power of two (extra queries are zero padding)
        scan_fn fn = pick_subranges(qbt, g->metric);
        if (!fn) return fail(FIR_ERR_ARG, "no sub-range kernel for qb=%d metric=%d", qbt, g->metric);
        hipLaunchKernelGGL(k_transpose_queries, dim3((unsigned)(((int64_t)kk * qbt + kBlock - 1) / kBlock)), dim3(kBlock), 0, st,
                           d_queries + (size_t)q0 * g->d, live, g->d, g->dp4, qbt, g->qt, (uint64_t*)nullptr, 0);
        const int max_waves = max_waves_for(g, fn, 0);
        ScanArgs a{};
        a.gal4 = g->gal4;
        a.qt = g->qt;
        a.n = g->n;
        a.tiles = (int32_t)g->tiles;
        a.dp4 = g->dp4;
        a.start = start;
        a.end = end;
        a.step = step;
        a.step2 = step2;
        a.waves = pick_waves(g->tiles, max_waves, g->cus * 4);
        a.row_offset = g->row_offset;
        a.out = d_out + (size_t)q0 * g->n;
        a.out_stride = g->n;
        a.nq = live;
        a.k = qb;
        a.nt = gallery_bytes(g) > kL2ResidentBytes ? 1 : 0;
        hipLaunchKernelGGL(fn, dim3(a.waves / 4), dim3(kBlock), 0, st, a);
        FIR_HIP(hipGetLastError());
    }
    return FIR_OK;
}

// hipGetDeviceCount, the first time on the private random state of fir_runtime_init_ (it starts the runtime).
hipError_t device_count(int* cnt) {
    static bool started = false;
    if (started) return hipGetDeviceCount(cnt);
    char state[256];
    char* caller = initstate(1u, state, sizeof state);
    const hipError_t e = hipGetDeviceCount(cnt);
    setstate(caller);
    started = true;
    return e;
}

int set_device(int device) {
    int cnt = 0;
    if (device_count(&cnt) != hipSuccess || cnt <= 0) return fail(FIR_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= cnt) return fail(FIR_ERR_NODEVICE, "device %d out of range (%d visible)", device, cnt);
    return fir_runtime_init_(device);
}

int gallery_alloc(int64_t n, int32_t d, int32_t metric, int32_t device, fir_gallery** out) {
    if (!out) return fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (n < 0 || d <= 0) return fail(FIR_ERR_ARG, "bad gallery shape n=%lld d=%d", (long long)n, d);
    if (n >= ((int64_t)1 << 31) - 64) return fail(FIR_ERR_ARG, "n=%lld does not fit 32-bit row indices", (long long)n);
    if (metric < 0 || metric > 2) return fail(FIR_ERR_ARG, "bad metric %d", metric);
    int rc = set_device(device);
    if (rc) return rc;
    hipDeviceProp_t prop;
    FIR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(FIR_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    fir_gallery* g = new (std::nothrow) fir_gallery();
    if (!g) return fail(FIR_ERR_NOMEM, "host allocation failed");
    g->device = device;
    g->cus = prop.multiProcessorCount;
    g->n = n;
    g->d = d;
    g->dp4 = (d + 3) / 4;
    g->tiles = (n + kTileRows - 1) / kTileRows;
    g->metric = metric;
    g->max_waves = g->cus * 8 * (kBlock / 64);
    hipError_t e;
    e = hipStreamCreateWithFlags(&g->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete g; return fail(FIR_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    const size_t f4 = (size_t)std::max<int64_t>(g->tiles, 1) * g->dp4 * 64;
    e = hipMalloc((void**)&g->gal4, f4 * sizeof(float4));
    if (e != hipSuccess) {
        (void)hipStreamDestroy(g->stream);
        delete g;
        return fail(FIR_ERR_NOMEM, "hipMalloc of %zu gallery bytes: %s", f4 * sizeof(float4), hipGetErrorString(e));
    }
    e = hipMalloc((void**)&g->range, 2 * sizeof(int32_t));
    if (e == hipSuccess) e = hipMemset(g->range, 0, 2 * sizeof(int32_t));
    if (e != hipSuccess) {
        (void)hipFree(g->gal4); (void)hipFree(g->range);
        (void)hipStreamDestroy(g->stream);
        delete g;
        return fail(FIR_ERR_NOMEM, "hipMalloc of the range flags: %s", hipGetErrorString(e));
    }
    *out = g;
    return FIR_OK;
}

int retile_slab(fir_gallery* g, const float* d_rows, int64_t slab_rows, int64_t row0, hipStream_t st) {
    const int64_t slab_tiles = (slab_rows + kTileRows - 1) / kTileRows;
    const int64_t total = slab_tiles * g->dp4 * 64;
    if (total == 0) return FIR_OK;
    const int64_t blocks = (total + kBlock - 1) / kBlock;
    if (blocks > 0x7FFFFFFF) return fail(FIR_ERR_ARG, "slab too large");
    hipLaunchKernelGGL(k_retile, dim3((unsigned)blocks), dim3(kBlock), 0, st, d_rows, slab_rows, row0, g->n, g->d, g->dp4, g->gal4,
                       g->range);
    FIR_HIP(hipGetLastError());
    return FIR_OK;
}

}  // namespace

extern "C" {

const char* fir_last_error(void) { return g_err; }
int fir_gallery_view_(fir_gallery* g, fir_gallery_view* out) {
    if (!g || !out) return FIR_ERR_ARG;
    out->device = g->device; out->cus = g->cus; out->n = g->n; out->d = g->d; out->metric = g->metric; out->row_offset = g->row_offset;
    out->cls = g->cls; out->stream = g->stream;
    return FIR_OK;
}
__global__ void k_runtime_warmup(int* p) {
    if (p && threadIdx.x == 0 && blockIdx.x == 0) *p = 1;
}
int fir_runtime_init_(int device) {
    static bool touched[64] = {};
    if (device >= 0 && device < 64 && !touched[device]) {
        char state[256];
        char* caller = initstate(1u, state, sizeof state);      // the runtime's start-up draws from here ...
        hipError_t e = hipSetDevice(device);
        void* p = nullptr;
        hipStream_t st = nullptr;
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) e = hipMalloc(&p, 256);             // forces the context ...
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (e == hipSuccess) {                                   // ... the code object load, a queue, a copy in each direction
            int h = 0;
            hipLaunchKernelGGL(k_runtime_warmup, dim3(1), dim3(64), 0, st, (int*)p);
            e = hipMemcpyAsync(&h, p, sizeof h, hipMemcpyDeviceToHost, st);
            if (e == hipSuccess) e = hipMemcpyAsync(p, &h, sizeof h, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
        }
        if (st) (void)hipStreamDestroy(st);
        if (p) (void)hipFree(p);
        setstate(caller);                                        // ... and the caller's rand() stream continues where it was
        if (e != hipSuccess) return fail(FIR_ERR_HIP, "device %d start-up failed: %s", device, hipGetErrorString(e));
        touched[device] = true;
        return FIR_OK;
    }
    FIR_HIP(hipSetDevice(device));
    return FIR_OK;
}
int fir_gallery_scratch_(fir_gallery* g, int slot, size_t bytes, void** out) {
    if (!g || !out || slot < 0 || slot >= 24) return fail(FIR_ERR_ARG, "bad scratch request");
    if (bytes > g->scratch_cap[slot]) {
        if (g->scratch[slot]) FIR_HIP(hipFree(g->scratch[slot]));
        g->scratch[slot] = nullptr;
        g->scratch_cap[slot] = 0;
        const size_t want = std::max<size_t>(bytes + bytes / 4, 4096);
        FIR_HIP(hipMalloc(&g->scratch[slot], want));
        FIR_HIP(hipMemset(g->scratch[slot], 0, want));        // a fresh slot reads as zeros (fir_rows_distances' arrival counter)
        g->scratch_cap[slot] = want;
    }
    *out = g->scratch[slot];
    return FIR_OK;
}
int fir_subrange_distances_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t step, float* d_out,
                                void* stream) {
    if (!g || !d_queries || !d_out) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb <= 0) return fail(FIR_ERR_ARG, "qb=%d must be positive", qb);
    if (step <= 0 || start < 0 || end > g->d || start >= end || (end - start) % step != 0)
        return fail(FIR_ERR_ARG, "sub-ranges of %d features do not tile [%d,%d) inside [0,%d)", step, start, end, g->d);
    if (g->n == 0) return FIR_OK;
    FIR_HIP(hipSetDevice(g->device));
    return subranges_dev(g, d_queries, qb, start, end, step, d_out, stream ? (hipStream_t)stream : g->stream);
}
int fir_split_distances_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t split, int32_t end, float* d_out, void* stream) {
    if (!g || !d_queries || !d_out) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb <= 0) return fail(FIR_ERR_ARG, "qb=%d must be positive", qb);
    if (split <= 0 || split >= end || end > g->d) return fail(FIR_ERR_ARG, "split %d / end %d outside (0,%d]", split, end, g->d);
    if (g->n == 0) return FIR_OK;
    FIR_HIP(hipSetDevice(g->device));
    hipStream_t st = stream ? (hipStream_t)stream : g->stream;
    if (split % (4 * kUSub) == 0 && (end - split) % (4 * kUSub) == 0) return subranges_dev(g, d_queries, qb, 0, end, split, d_out, st, end - split);
    const int rc = range_dev(g, d_queries, qb, 0, split, d_out, st);
    if (rc) return rc;
    return range_dev(g, d_queries, qb, split, end, d_out + (size_t)qb * g->n, st);
}
int fir_gallery_tiled_(fir_gallery* g, const void** gal4, int* dp4) {
    if (!g || !gal4 || !dp4) return FIR_ERR_ARG;
    *gal4 = g->gal4;
    *dp4 = g->dp4;
    return FIR_OK;
}
// internal: lets the library's other translation units (fir_cls.hip) report through fir_last_error()
void fir_set_last_error_(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
int fir_version(void) { return 100; }

int fir_device_count(void) {
    int cnt = 0;
    if (device_count(&cnt) != hipSuccess) return 0;
    return cnt;
}

int fir_device_info(int32_t device, char* name, int32_t cap, int32_t* cus, int64_t* hbm_bytes) {
    int cnt = 0;
    if (device_count(&cnt) != hipSuccess || device < 0 || device >= cnt)
        return fail(FIR_ERR_NODEVICE, "device %d not available", device);
    hipDeviceProp_t prop;
    FIR_HIP(hipGetDeviceProperties(&prop, device));
    if (name && cap > 0) { strncpy(name, prop.gcnArchName, (size_t)cap - 1); name[cap - 1] = 0; }
    if (cus) *cus = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)prop.totalGlobalMem;
    return FIR_OK;
}

int fir_device_peak_hbm_gbs(int32_t device, double* gbs) {
    int cnt = 0;
    if (!gbs) return fail(FIR_ERR_ARG, "gbs is NULL");
    if (device_count(&cnt) != hipSuccess || device < 0 || device >= cnt)
        return fail(FIR_ERR_NODEVICE, "device %d not available", device);
    hipDeviceProp_t prop;
    FIR_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return fail(FIR_ERR_NODEVICE, "device %d is %s, not gfx950", device, prop.gcnArchName);
    *gbs = 8000.0;   // MI350X / MI355X: 8 stacks of HBM3E, 8 TB/s (the runtime's clock x bus-width fields do not give this)
    return FIR_OK;
}

int fir_gallery_create(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric, int32_t device,
                       fir_gallery** out) {
    if (n > 0 && !rows) return fail(FIR_ERR_ARG, "rows is NULL");
    fir_gallery* g = nullptr;
    int rc = gallery_alloc(n, d, metric, device, &g);
    if (rc) return rc;
    // upload in slabs of whole tiles (<= 256 MiB of rows each) and re-tile on the device
    int64_t slab = std::max<int64_t>(kTileRows, ((int64_t)(256u << 20) / ((int64_t)d * 4)) / kTileRows * kTileRows);
    slab = std::min<int64_t>(slab, std::max<int64_t>(g->tiles, 1) * kTileRows);
    float* stage = nullptr;
    hipError_t e = hipMalloc((void**)&stage, (size_t)slab * d * sizeof(float));
    if (e != hipSuccess) { fir_gallery_destroy(g); return fail(FIR_ERR_NOMEM, "staging hipMalloc: %s", hipGetErrorString(e)); }
    for (int64_t r0 = 0; r0 < std::max<int64_t>(g->tiles, 1) * kTileRows && rc == FIR_OK; r0 += slab) {
        const int64_t have = std::max<int64_t>(0, std::min<int64_t>(slab, n - r0));
        if (have > 0) {
            e = hipMemcpyAsync(stage, rows + r0 * d, (size_t)have * d * sizeof(float), hipMemcpyHostToDevice, g->stream);
            if (e != hipSuccess) { rc = fail(FIR_ERR_HIP, "gallery upload: %s", hipGetErrorString(e)); break; }
        }
        if (have > 0) rc = retile_slab(g, stage, have, r0, g->stream);
        e = hipStreamSynchronize(g->stream);   // the staging buffer is reused by the next slab
        if (e != hipSuccess && rc == FIR_OK) rc = fail(FIR_ERR_HIP, "gallery retile: %s", hipGetErrorString(e));
    }
    (void)hipFree(stage);
    if (rc == FIR_OK && class_no && n > 0) {
        e = hipMalloc((void**)&g->cls, (size_t)n * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemcpy(g->cls, class_no, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = fail(FIR_ERR_HIP, "class upload: %s", hipGetErrorString(e));
    }
    if (rc) { fir_gallery_destroy(g); return rc; }
    { int32_t r0 = 1; if (hipMemcpy(&r0, g->range, sizeof r0, hipMemcpyDeviceToHost) == hipSuccess) g->gallery_plain = r0 == 0; }
    *out = g;
    return FIR_OK;
}

int fir_gallery_create_dev(const float* d_rows, int64_t n, int32_t d, const int32_t* d_class_no, int32_t metric,
                           int32_t device, void* stream, fir_gallery** out) {
    if (n > 0 && !d_rows) return fail(FIR_ERR_ARG, "d_rows is NULL");
    fir_gallery* g = nullptr;
    int rc = gallery_alloc(n, d, metric, device, &g);
    if (rc) return rc;
    hipStream_t st = stream ? (hipStream_t)stream : g->stream;
    // slabs keep the launch grid inside 2^31 blocks
    const int64_t slab = std::max<int64_t>(kTileRows, ((int64_t)1 << 30) / ((int64_t)g->dp4 * 4) / kTileRows * kTileRows);
    for (int64_t r0 = 0; r0 < g->tiles * kTileRows && rc == FIR_OK; r0 += slab) {
        const int64_t span = std::min<int64_t>(slab, g->tiles * kTileRows - r0);
        rc = retile_slab(g, d_rows + r0 * d, span, r0, st);
    }
    if (rc == FIR_OK && d_class_no && n > 0) {
        hipError_t e = hipMalloc((void**)&g->cls, (size_t)n * sizeof(int32_t));
        if (e == hipSuccess) e = hipMemcpyAsync(g->cls, d_class_no, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) rc = fail(FIR_ERR_HIP, "class copy: %s", hipGetErrorString(e));
    }
    if (rc == FIR_OK) {
        hipError_t e = hipStreamSynchronize(st);
        if (e != hipSuccess) rc = fail(FIR_ERR_HIP, "gallery retile: %s", hipGetErrorString(e));
    }
    if (rc) { fir_gallery_destroy(g); return rc; }
    { int32_t r0 = 1; if (hipMemcpy(&r0, g->range, sizeof r0, hipMemcpyDeviceToHost) == hipSuccess) g->gallery_plain = r0 == 0; }
    *out = g;
    return FIR_OK;
}

int fir_gallery_destroy(fir_gallery* g) {
    if (!g) return FIR_OK;
    (void)hipSetDevice(g->device);
    if (g->stream) (void)hipStreamSynchronize(g->stream);
    if (g->gemm) { fir_gemm_destroy(g->gemm); g->gemm = nullptr; }
    for (auto& ps : g->gemm_prefix) if (ps.m) { fir_gemm_destroy(ps.m); ps.m = nullptr; ps.end = 0; }
    for (hipEvent_t e : g->ev) (void)hipEventDestroy(e);
    (void)hipFree(g->gal4); (void)hipFree(g->cls); (void)hipFree(g->qt); (void)hipFree(g->dq); (void)hipFree(g->dkeys);
    (void)hipFree(g->part); (void)hipFree(g->dout); (void)hipFree(g->didx); (void)hipFree(g->range); (void)hipFree(g->one_keys); (void)hipFree(g->rowsum);
    if (g->pin) (void)hipHostFree(g->pin);
    for (void* p : g->scratch) if (p) (void)hipFree(p);
    if (g->stream) (void)hipStreamDestroy(g->stream);
    delete g;
    return FIR_OK;
}

int fir_gallery_info(const fir_gallery* g, int64_t* n, int32_t* d, int32_t* metric, int32_t* device) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (n) *n = g->n;
    if (d) *d = g->d;
    if (metric) *metric = g->metric;
    if (device) *device = g->device;
    return FIR_OK;
}

int fir_gallery_set_metric(fir_gallery* g, int32_t metric) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (metric < 0 || metric > 2) return fail(FIR_ERR_ARG, "bad metric %d", metric);
    g->metric = metric;
    return FIR_OK;
}

int fir_gallery_set_large_batch_mfma(fir_gallery* g, int32_t min_queries) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    g->large_batch_min = min_queries < 0 ? -1 : min_queries;
    g->gemm_failed = false;
    if (min_queries == 0 && g->gemm) { fir_gemm_destroy(g->gemm); g->gemm = nullptr; }
    if (min_queries == 0) for (auto& ps : g->gemm_prefix) if (ps.m) { fir_gemm_destroy(ps.m); ps.m = nullptr; ps.end = 0; }
    return FIR_OK;
}

int fir_gallery_mfma_stats_ex(fir_gallery* g, int64_t out[3]) {
    if (!g || !out) return fail(FIR_ERR_ARG, "NULL argument");
    int64_t o[3];
    out[0] = out[1] = out[2] = 0;
    if (g->gemm && fir_gemm_stats_ex(g->gemm, o) == FIR_OK) { out[0] += o[0]; out[1] += o[1]; out[2] += o[2]; }
    for (auto& ps : g->gemm_prefix)
        if (ps.m && fir_gemm_stats_ex(ps.m, o) == FIR_OK) { out[0] += o[0]; out[1] += o[1]; out[2] += o[2]; }
    // (states dropped since -- fir_gallery_set_large_batch_mfma(g, 0) frees them -- took their counters along)
    return FIR_OK;
}

int fir_gallery_mfma_uncertified_notes(fir_gallery* g, float out[32], int32_t* count) {
    if (!g || !out || !count) return fail(FIR_ERR_ARG, "NULL argument");
    *count = 0;
    if (!g->gemm) return FIR_OK;
    return fir_gemm_uncertified_notes(g->gemm, out, count);
}

int fir_gallery_mfma_stats(fir_gallery* g, int64_t* passes, int64_t* fallback_queries) {
    int64_t o[3];
    const int rc = fir_gallery_mfma_stats_ex(g, o);
    if (rc) return rc;
    if (passes) *passes = o[0];
    if (fallback_queries) *fallback_queries = o[2];
    return FIR_OK;
}

int fir_gallery_memory_bytes(fir_gallery* g, int64_t* tiled, int64_t* fp16_fragments, int64_t* rowmajor_shadow, int64_t* scratch) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    int64_t fr = 0, rm = 0, sc = 0, a = 0, b = 0, c = 0;
    if (g->gemm) { fir_gemm_memory_bytes_(g->gemm, &a, &b, &c); fr += a; rm += b; sc += c; }
    for (auto& ps : g->gemm_prefix)
        if (ps.m) { fir_gemm_memory_bytes_(ps.m, &a, &b, &c); fr += a; rm += b; sc += c; }
    sc += (int64_t)(g->qt_cap * sizeof(float) + g->dq_cap * sizeof(float) + g->dkeys_cap * sizeof(uint64_t) + g->part_cap * sizeof(uint64_t) +
                    g->dout_cap * sizeof(float) + g->didx_cap * sizeof(int32_t));
    for (size_t i = 0; i < 24; ++i) sc += (int64_t)g->scratch_cap[i];
    sc += (int64_t)(g->rowsum_cap * sizeof(float));
    if (tiled) *tiled = (int64_t)g->tiles * fir::kTileRows * (int64_t)((g->d + 3) / 4) * 16 + (g->cls ? (int64_t)g->n * 4 : 0);
    if (fp16_fragments) *fp16_fragments = fr;
    if (rowmajor_shadow) *rowmajor_shadow = rm;
    if (scratch) *scratch = sc;
    return FIR_OK;
}

int fir_gallery_set_shadow_copies(fir_gallery* g, int32_t mode) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (mode != FIR_SHADOW_NONE && mode != FIR_SHADOW_FP16 && mode != FIR_SHADOW_ALL) return fail(FIR_ERR_ARG, "shadow mode %d", mode);
    if (mode != g->shadow_mode) {
        if (g->gemm) { fir_gemm_destroy(g->gemm); g->gemm = nullptr; }
        for (auto& ps : g->gemm_prefix) if (ps.m) { fir_gemm_destroy(ps.m); ps.m = nullptr; ps.end = 0; }
        g->gemm_failed = false;
    }
    g->shadow_mode = mode;
    return FIR_OK;
}

int fir_gallery_set_row_offset(fir_gallery* g, int64_t first_global_row) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (first_global_row < 0 || first_global_row + g->n >= ((int64_t)1 << 31))
        return fail(FIR_ERR_ARG, "row offset %lld + n does not fit 32-bit indices", (long long)first_global_row);
    g->row_offset = first_global_row;
    if (g->gemm) { fir_gemm_destroy(g->gemm); g->gemm = nullptr; }   // it caches the offset; rebuilt on next use
    for (auto& ps : g->gemm_prefix) if (ps.m) { fir_gemm_destroy(ps.m); ps.m = nullptr; ps.end = 0; }
    return FIR_OK;
}

int fir_feature_distance(const float* lhs, const float* rhs, int32_t len, int32_t start_pos, int32_t end_pos,
                         int32_t metric, int32_t device, float* out) {
    if (!lhs || !rhs || !out) return fail(FIR_ERR_ARG, "NULL argument");
    if (start_pos < 0 || end_pos > len || start_pos >= end_pos) return fail(FIR_ERR_ARG, "bad range [%d,%d) of %d", start_pos, end_pos, len);
    if (metric < 0 || metric > 2) return fail(FIR_ERR_ARG, "bad metric %d", metric);
    int rc = set_device(device);
    if (rc) return rc;
    float* buf = nullptr;
    FIR_HIP(hipMalloc((void**)&buf, (size_t)(2 * len + 1) * sizeof(float)));
    hipError_t e = hipMemcpy(buf, lhs, (size_t)len * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(buf + len, rhs, (size_t)len * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        if (metric == 0) hipLaunchKernelGGL(k_pair_distance<0>, dim3(1), dim3(64), 0, 0, buf, buf + len, start_pos, end_pos, buf + 2 * len);
        else if (metric == 1) hipLaunchKernelGGL(k_pair_distance<1>, dim3(1), dim3(64), 0, 0, buf, buf + len, start_pos, end_pos, buf + 2 * len);
        else hipLaunchKernelGGL(k_pair_distance<2>, dim3(1), dim3(64), 0, 0, buf, buf + len, start_pos, end_pos, buf + 2 * len);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, buf + 2 * len, sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(buf);
    if (e != hipSuccess) return fail(FIR_ERR_HIP, "pair distance: %s", hipGetErrorString(e));
    return FIR_OK;
}

namespace {
// 0 = *m is the handle's fir_gemm for features [0, end), 1 = this shape stays with the scan, < 0 = error
int ensure_gemm(fir_gallery* g, int32_t end, fir_gemm** m, int* warm = nullptr, int warm_calls = 0) {
    // warm / warm_calls (automatic dispatch of cases that save tens to hundreds of microseconds per call: cache-resident galleries,
    // one-query calls): the state -- an fp16 copy, scratch, and the first time in a process ~12 ms of kernel loading -- is only built for
    // a gallery that keeps getting such calls; the first warm_calls of them take the scan (a test harness that makes ONE batched call
    // against a 3 030-row gallery would pay 16 ms to save 2)
    const bool whole = end == g->d;
    if (g->shadow_mode == FIR_SHADOW_NONE && g->large_batch_min <= 0) return 1;     // the caller forbade every copy: the scan
    fir_gallery::PrefixSlot* ps = nullptr;
    if (!whole) {
        for (auto& c : g->gemm_prefix) if (c.m && c.end == end) ps = &c;
        if (!ps) {                                                  // the free slot, else the one used longest ago
            ps = &g->gemm_prefix[0];
            for (auto& c : g->gemm_prefix) if (!c.m) { ps = &c; break; } else if (c.used < ps->used) ps = &c;
        }
    }
    fir_gemm*& slot = whole ? g->gemm : ps->m;
    const bool have = slot && (whole || ps->end == end);
    if (warm && !have && g->large_batch_min < 0) {
        ++*warm;
        g->warm_left = *warm <= warm_calls ? warm_calls - *warm + 1 : 0;
        if (*warm <= warm_calls) return 1;
    }
    if (!whole && slot && ps->end != end) {                         // a third prefix: its fragments replace the least recently used ones
        fir_gemm_destroy(slot);
        slot = nullptr;
        ps->end = 0;
    }
    if (!slot) {
        const int rc = fir_gemm_create_range_ex_(g, FIR_GEMM_F16, whole ? 0 : end, g->shadow_mode == FIR_SHADOW_FP16 ? 0 : -1, &slot);
        if (rc) {
            slot = nullptr;
            if (g->large_batch_min > 0 || (rc != FIR_ERR_ARG && rc != FIR_ERR_NOMEM)) return rc;    // asked for explicitly, or a real failure
            g->gemm_failed = true;                                           // automatic: this shape (or this much HBM) stays with the scan
            return 1;
        }
        if (!whole) ps->end = end;
    }
    if (!whole) ps->used = ++g->prefix_clock;
    *m = slot;
    return 0;
}
// The matrix-core path for this call, if it applies: 0 = done, 1 = take the scan, < 0 = error.
int try_mfma(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, uint64_t* d_keys, hipStream_t st,
             const float* h_queries = nullptr) {
    // h_queries: the queries are still on the host and d_queries is the (writable) device buffer they are staged through
    if (!h_queries && wants_few(g, qb, start, end)) {
        fir_gemm* mf = nullptr;
        const int rcf = ensure_gemm(g, end, &mf, &g->few_hits, 16);
        if (rcf) return rcf;
        const int rcs = fir_gemm_search_few_keys_dev(mf, d_queries, qb, d_keys, st);
        if (rcs != FIR_ERR_NOMEM) return rcs;
        g->few_hits = -(1 << 30);          // no room for the proxy table: this gallery's one-query calls stay with the exact scan
        return 1;
    }
    if (!wants_mfma(g, qb, start, end)) return 1;
    fir_gemm* m = nullptr;
    const int rc = g->n < kAutoMfmaRows ? ensure_gemm(g, end, &m, &g->small_hits, 3) : ensure_gemm(g, end, &m);
    if (rc) return rc;
    const int rcs = h_queries ? fir_gemm_search_staged_(m, h_queries, (float*)d_queries, qb, 1, d_keys, st)
                              : fir_gemm_search_top1_keys_dev(m, d_queries, qb, d_keys, st);
    if (rcs == FIR_ERR_NOMEM && g->large_batch_min < 0) {       // no room for this call's candidate lists: the exact scan answers (automatic mode)
        (void)hipGetLastError();
        return 1;
    }
    return rcs;
}
int try_mfma_topk(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, uint64_t* d_keys, hipStream_t st) {
    if (k < 2 || !wants_mfma(g, qb, start, end)) return 1;       // K = 1 callers use the top-1 entry points
    fir_gemm* m = nullptr;
    const int rc = g->n < kAutoMfmaRows ? ensure_gemm(g, end, &m, &g->small_hits, 3) : ensure_gemm(g, end, &m);
    if (rc) return rc;
    const int rcs = fir_gemm_search_topk_keys_dev(m, d_queries, qb, k, d_keys, st);
    if (rcs == FIR_ERR_NOMEM && g->large_batch_min < 0) {
        (void)hipGetLastError();
        return 1;
    }
    return rcs;
}
}  // namespace

int fir_search_topk_exact_keys_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t end_pos, int32_t k, uint64_t* d_keys, void* stream) {
    if (!g || !d_keys || (qb > 0 && !d_queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb <= 0) return qb < 0 ? fail(FIR_ERR_ARG, "qb < 0") : FIR_OK;
    if (k < 1 || k > kKMax) return fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kKMax);
    int32_t start_pos = 0;
    const int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    const fir_dispatch_info saved = g->last;          // the caller's dispatch record stays the matrix-core pass's, not this fallback's
    const int rc2 = topk_dev(g, d_queries, qb, 0, end_pos, k, d_keys, stream ? (hipStream_t)stream : g->stream, true, false);
    g->last = saved;
    return rc2;
}

int fir_search_top1_keys_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                             uint64_t* d_keys, void* stream) {
    if (!g || !d_keys || (qb > 0 && !d_queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    hipStream_t st = stream ? (hipStream_t)stream : g->stream;
    g->call_launches = 0;
    g->warm_left = 0;
    rc = try_mfma(g, d_queries, qb, start_pos, end_pos, d_keys, st);
    if (rc <= 0) return rc;
    return top1_dev(g, d_queries, qb, start_pos, end_pos, d_keys, st);
}

// The exact streaming scan, whatever the batch size (fir_gemm.hip sends its uncertified queries here).
int fir_search_top1_exact_keys_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos, uint64_t* d_keys,
                                    void* stream) {
    if (!g || !d_keys || (qb > 0 && !d_queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb <= 0) return qb < 0 ? fail(FIR_ERR_ARG, "qb < 0") : FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    const bool was_profiling = g->profiling;
    g->profiling = false;
    g->quiet = true;
    rc = top1_dev(g, d_queries, qb, start_pos, end_pos, d_keys, stream ? (hipStream_t)stream : g->stream);
    g->quiet = false;
    g->profiling = was_profiling;
    return rc;
}

// Profiling hooks for the library's other translation units: an event pair around one launch on `st`.
int fir_gallery_profile_begin_(fir_gallery* g, void* st) {
    if (!g->profiling) return FIR_OK;
    if (g->ev_used + 2 > g->ev.size())
        for (int i = 0; i < 64; ++i) {
            hipEvent_t e;
            FIR_HIP(hipEventCreate(&e));
            g->ev.push_back(e);
        }
    FIR_HIP(hipEventRecord(g->ev[g->ev_used], (hipStream_t)st));
    return FIR_OK;
}
int fir_gallery_profile_end_(fir_gallery* g, void* st, double bytes_alg) {
    if (!g->profiling) return FIR_OK;
    FIR_HIP(hipEventRecord(g->ev[g->ev_used + 1], (hipStream_t)st));
    g->ev_used += 2;
    g->last_bytes = bytes_alg;
    return FIR_OK;
}
void fir_gallery_note_dispatch_(fir_gallery* g, const void* fn, const char* name, int first, int gx, int gy, int block, size_t dyn_lds, int qpp,
                                double bytes, double flops) {
    note_dispatch(g, fn, name, first ? 0 : 1, gx, gy, block, dyn_lds, qpp, bytes, flops, 1);
}

namespace {
std::mutex g_knob_mu;
std::vector<std::string> g_knobs;        // names of the set FIR_* variables the library has looked at, in order of first use
void knobs_list_(char* out, size_t cap) {
    std::lock_guard<std::mutex> lk(g_knob_mu);
    size_t used = 0;
    out[0] = 0;
    for (const std::string& k : g_knobs) {
        if (used + k.size() + 6 >= cap) { std::snprintf(out + used, cap - used, "..."); return; }
        used += (size_t)std::snprintf(out + used, cap - used, "%s%s", used ? " " : "", k.c_str());
    }
}
}  // namespace

extern "C" const char* fir_knob_(const char* name) {
    const char* v = std::getenv(name);
    if (v) {
        std::lock_guard<std::mutex> lk(g_knob_mu);
        if (std::find(g_knobs.begin(), g_knobs.end(), name) == g_knobs.end()) g_knobs.emplace_back(name);
    }
    return v;
}

int fir_gallery_last_dispatch(fir_gallery* g, fir_dispatch_info* out) {
    if (!g || !out) return fail(FIR_ERR_ARG, "NULL argument");
    if (out->struct_bytes < 8 || out->struct_bytes > (int32_t)sizeof(fir_dispatch_info)) return fail(FIR_ERR_ARG, "fir_dispatch_info.struct_bytes = %d", out->struct_bytes);
    const int32_t nb = out->struct_bytes;
    fir_dispatch_info tmp = g->last;
    tmp.warmup_calls_left = g->warm_left;
    knobs_list_(tmp.knobs, sizeof tmp.knobs);
    tmp.struct_bytes = nb;
    std::memcpy(out, &tmp, (size_t)nb);
    return FIR_OK;
}

namespace {
constexpr size_t kPinQueryBytes = 512 * 1024;    // host-pointer calls up to this many query bytes take the pinned path (a 64 x 1536 TWD batch fits)
constexpr size_t kPinKeys = 4096;                // and up to this many result keys
// ticket != 0 (single-block launches only): after the keys, host_keys[n] <- ticket -- the host spins on that word instead of
// synchronising the stream (wait_ticket)
__global__ void __launch_bounds__(kBlock) k_publish_keys(const uint64_t* __restrict__ keys, int n, uint64_t* __restrict__ host_keys,
                                                         uint64_t ticket) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) host_keys[i] = keys[i];
    if (ticket) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(host_keys + n, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// Spin (2 ms at most, then the stream synchronisation) until the kernels queued on the handle's stream have written
// `ticket` to the pinned word: cheaper than hipStreamSynchronize for calls that take tens of microseconds.
int wait_ticket(fir_gallery* g, volatile uint64_t* flag, uint64_t ticket) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket; ++spins) {
        if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
            FIR_HIP(hipStreamSynchronize(g->stream));       // a long or failed launch: let the runtime report it
            if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket) return fail(FIR_ERR_HIP, "the result ticket was not published");
            break;
        }
    }
    return FIR_OK;
}
// One L2 query against a gallery of few tiles -- the reference's own call pattern, recognize() per test image against
// ~3 000 rows. Such a call is mostly fixed cost (three launches and a stream synchronisation next to ~20 us of kernels),
// so here it is ONE launch and no synchronisation: the scan stages the query in LDS straight from the pinned buffer (a
// one-query tile needs no transposition), its last workgroup writes the key and then a ticket to pinned host memory and
// re-arms the device key, and the host spins on the ticket (falling back to hipStreamSynchronize after 2 ms).
// 3 030 x 1536: 34 instead of 40 us per call (profiles/r01_sweep_notes.md). Returns 1 when the shape does not qualify.
int top1_one_query(fir_gallery* g, const float* pinned_query, int32_t start, int32_t end, uint64_t* pinned_key) {
    static const bool off = fir_knob_("FIR_NO_ONE_QUERY") != nullptr;      // experiments
    // up to 256 tiles (16 384 rows): every workgroup of this form ends on a device-wide fence before it is counted, and beyond
    // ~90 workgroups those fences cost more than the two extra launches of the general path (12 000 x 512: 25.5 against
    // 28.1 us; 24 000: 32.8 / 32.2; 50 000: 44.4 / 35.4)
    if (off || g->metric != kL2 || g->n <= 0 || g->tiles > 256 || g->tiles_limit > 0 || g->profiling) return 1;
    size_t lds_bytes = 0;
    scan_fn fn = pick_deep(kEpiTop1, 1, g->metric, g->dp4, &lds_bytes);
    if (!fn) return 1;
    if (!g->one_keys) {
        FIR_HIP(hipMalloc((void**)&g->one_keys, 64));
        g->one_done = (int32_t*)(g->one_keys + 4);
        const uint64_t init[8] = {kKeyNone, kKeyNone, kKeyNone, kKeyNone, 0, 0, 0, 0};
        FIR_HIP(hipMemcpy(g->one_keys, init, sizeof init, hipMemcpyHostToDevice));
    }
    const int max_waves = max_waves_for(g, fn, lds_bytes);
    const int waves = g->waves_req > 0 ? std::min(g->waves_req, max_waves) : pick_waves(g->tiles, max_waves, g->cus * 4);
    g->last_waves = waves;
    ScanArgs a{};
    a.qt = pinned_query;
    a.gal4 = g->gal4;
    a.row_offset = g->row_offset;
    a.n = g->n;
    a.tiles = (int32_t)g->tiles;
    a.dp4 = g->dp4;
    a.start = start;
    a.end = end;
    a.waves = waves;
    a.keys = g->one_keys;
    a.nq = 1;
    a.qt_stride = (int64_t)g->dp4 * 4;
    a.nt = gallery_bytes(g) > kL2ResidentBytes ? 1 : 0;
    a.publish = pinned_key;
    a.done = g->one_done;
    a.ticket = ++g->one_ticket;
    hipLaunchKernelGGL(fn, dim3(waves / 4, 1), dim3(kBlock), lds_bytes, g->stream, a);
    const hipError_t le = hipGetLastError();
    if (le == hipSuccess && wait_ticket(g, pinned_key + 1, a.ticket) == FIR_OK) return FIR_OK;
    // The launch failed or its last workgroup never published: only that workgroup re-arms the device key and the arrival
    // counter, so they are re-armed here (else every later one-query call would wait 2 ms and fail), and this call goes
    // through the general path. If the device itself is gone that path reports it.
    (void)hipStreamSynchronize(g->stream);
    const uint64_t init[8] = {kKeyNone, kKeyNone, kKeyNone, kKeyNone, 0, 0, 0, 0};
    (void)hipMemcpy(g->one_keys, init, sizeof init, hipMemcpyHostToDevice);
    return 1;
}

int ensure_pin(fir_gallery* g) {
    if (g->pin) return FIR_OK;
    FIR_HIP(hipHostMalloc(&g->pin, kPinQueryBytes + kPinKeys * sizeof(uint64_t), hipHostMallocDefault));
    return FIR_OK;
}
}  // namespace

int fir_gallery_pin_(fir_gallery* g, void** base, size_t* query_bytes, uint64_t** results) {
    if (!g) return FIR_ERR_ARG;
    const int rc = ensure_pin(g);
    if (rc) return rc;
    *base = g->pin;
    *query_bytes = kPinQueryBytes;
    *results = (uint64_t*)((char*)g->pin + kPinQueryBytes);
    return FIR_OK;
}
uint64_t fir_gallery_next_ticket_(fir_gallery* g) { return ++g->one_ticket; }
uint64_t fir_gallery_next_counter_(fir_gallery* g, int slot) { return g->counters[slot & 3]++; }
int fir_gallery_wait_ticket_(fir_gallery* g, volatile uint64_t* flag, uint64_t ticket) { return wait_ticket(g, flag, ticket); }

int fir_search_top1(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t* idx,
                    float* dist) {
    if (!g || (qb > 0 && !queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    const bool mfma = wants_mfma(g, qb, start_pos, end_pos);
    g->call_launches = 0;
    if (!mfma && (size_t)qb * g->d * sizeof(float) <= kPinQueryBytes && (size_t)qb <= kPinKeys) {
        // small call: queries are read from, and keys written to, pinned host memory by the kernels themselves
        if ((rc = ensure_pin(g))) return rc;
        if ((rc = grow(g->dkeys, g->dkeys_cap, (size_t)qb))) return rc;
        float* hq = (float*)g->pin;
        uint64_t* hk = (uint64_t*)((char*)g->pin + kPinQueryBytes);
        std::memcpy(hq, queries, (size_t)qb * g->d * sizeof(float));
        if (qb == 1) {
            for (int k = g->d; k < g->dp4 * 4; ++k) hq[k] = 0.0f;        // the one-query tile, zero padded
            rc = top1_one_query(g, hq, start_pos, end_pos, hk);
            if (rc < 0) return rc;
            if (rc == FIR_OK) return fir_keys_unpack(hk, 1, idx, dist);
        }
        rc = try_mfma(g, hq, qb, start_pos, end_pos, g->dkeys, g->stream);              // (the few-query form on large galleries; else 1)
        if (rc < 0) return rc;
        if (rc > 0 && (rc = top1_dev(g, hq, qb, start_pos, end_pos, g->dkeys, g->stream))) return rc;
        const uint64_t ticket = qb <= kBlock && !g->profiling ? ++g->one_ticket : 0;     // one block publishes: the host can wait on its ticket
        hipLaunchKernelGGL(k_publish_keys, dim3((qb + kBlock - 1) / kBlock), dim3(kBlock), 0, g->stream, g->dkeys, qb, hk, ticket);
        FIR_HIP(hipGetLastError());
        if (ticket) { if ((rc = wait_ticket(g, hk + qb, ticket))) return rc; }
        else FIR_HIP(hipStreamSynchronize(g->stream));
        return fir_keys_unpack(hk, qb, idx, dist);
    }
    if ((rc = grow(g->dq, g->dq_cap, (size_t)qb * g->d))) return rc;
    if ((rc = grow(g->dkeys, g->dkeys_cap, (size_t)qb))) return rc;
    // large L2 batches go through the matrix cores (same keys, fir_gemm.hip) unless switched off; their queries are uploaded
    // super-batch by super-batch under the passes of the one before
    rc = try_mfma(g, g->dq, qb, start_pos, end_pos, g->dkeys, g->stream, queries);
    if (rc > 0) {
        FIR_HIP(hipMemcpyAsync(g->dq, queries, (size_t)qb * g->d * sizeof(float), hipMemcpyHostToDevice, g->stream));
        rc = top1_dev(g, g->dq, qb, start_pos, end_pos, g->dkeys, g->stream);
    }
    if (rc) return rc;
    std::vector<uint64_t> keys((size_t)qb);
    FIR_HIP(hipMemcpyAsync(keys.data(), g->dkeys, (size_t)qb * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    FIR_HIP(hipStreamSynchronize(g->stream));
    return fir_keys_unpack(keys.data(), qb, idx, dist);
}

uint64_t fir_key_pack(float dist, int32_t idx) {
    if (idx < 0) return kKeyNone;
    return key_pack(dist, (uint32_t)idx);
}

int fir_keys_unpack(const uint64_t* keys, int32_t n, int32_t* idx, float* dist) {
    if (!keys && n > 0) return fail(FIR_ERR_ARG, "keys is NULL");
    for (int32_t i = 0; i < n; ++i) {
        if (keys[i] == kKeyNone) {
            if (idx) idx[i] = -1;
            if (dist) dist[i] = kNotFound;
        } else {
            if (idx) idx[i] = (int32_t)(uint32_t)(keys[i] & 0xFFFFFFFFull);
            if (dist) dist[i] = f32_from_orderable((uint32_t)(keys[i] >> 32));
        }
    }
    return FIR_OK;
}

int fir_gallery_classes_of(fir_gallery* g, const int32_t* idx, int32_t n, int32_t* class_out) {
    if (!g || !idx || !class_out) return fail(FIR_ERR_ARG, "NULL argument");
    if (!g->cls) return fail(FIR_ERR_STATE, "gallery was created without class labels");
    if (n <= 0) return FIR_OK;
    FIR_HIP(hipSetDevice(g->device));
    int rc = grow(g->didx, g->didx_cap, (size_t)2 * n);
    if (rc) return rc;
    FIR_HIP(hipMemcpyAsync(g->didx, idx, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, g->stream));
    hipLaunchKernelGGL(k_classes_of, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, g->stream, g->cls, g->n, g->row_offset,
                       g->didx, n, g->didx + n);
    FIR_HIP(hipGetLastError());
    FIR_HIP(hipMemcpyAsync(class_out, g->didx + n, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, g->stream));
    FIR_HIP(hipStreamSynchronize(g->stream));
    return FIR_OK;
}

int fir_search_topk_keys_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                             int32_t k, uint64_t* d_keys, void* stream) {
    if (!g || !d_keys || (qb > 0 && !d_queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > kKMax) return fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kKMax);
    if (qb == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    return topk_dev(g, d_queries, qb, start_pos, end_pos, k, d_keys, stream ? (hipStream_t)stream : g->stream);
}

int fir_search_topk(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t k,
                    int32_t* idx, float* dist) {
    if (!g || (qb > 0 && !queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > kKMax) return fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kKMax);
    if (qb == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    if ((rc = grow(g->dkeys, g->dkeys_cap, (size_t)qb * k))) return rc;
    if ((size_t)qb * g->d * sizeof(float) <= kPinQueryBytes && qb * k <= kBlock && qb < 8 && !g->profiling) {
        // small call (below the candidate-list form, which synchronises by itself): pinned queries in, keys + ticket out, as in fir_search_top1
        if ((rc = ensure_pin(g))) return rc;
        float* hq = (float*)g->pin;
        uint64_t* hk = (uint64_t*)((char*)g->pin + kPinQueryBytes);
        std::memcpy(hq, queries, (size_t)qb * g->d * sizeof(float));
        if ((rc = topk_dev(g, hq, qb, start_pos, end_pos, k, g->dkeys, g->stream))) return rc;
        const uint64_t ticket = ++g->one_ticket;
        hipLaunchKernelGGL(k_publish_keys, dim3(1), dim3(kBlock), 0, g->stream, g->dkeys, qb * k, hk, ticket);
        FIR_HIP(hipGetLastError());
        if ((rc = wait_ticket(g, hk + qb * k, ticket))) return rc;
        return fir_keys_unpack(hk, qb * k, idx, dist);
    }
    if ((rc = grow(g->dq, g->dq_cap, (size_t)qb * g->d))) return rc;
    // large L2 batches: the matrix cores, the queries uploaded super-batch by super-batch under the passes (as fir_search_top1)
    rc = 1;
    if (k >= 2 && wants_mfma(g, qb, start_pos, end_pos)) {
        fir_gemm* m = nullptr;
        rc = g->n < kAutoMfmaRows ? ensure_gemm(g, end_pos, &m, &g->small_hits, 3) : ensure_gemm(g, end_pos, &m);
        if (rc == 0) rc = fir_gemm_search_staged_(m, queries, g->dq, qb, k, g->dkeys, g->stream);
        if (rc < 0) return rc;
    }
    if (rc > 0) {
        FIR_HIP(hipMemcpyAsync(g->dq, queries, (size_t)qb * g->d * sizeof(float), hipMemcpyHostToDevice, g->stream));
        if ((rc = topk_dev(g, g->dq, qb, start_pos, end_pos, k, g->dkeys, g->stream, true, false))) return rc;   // (the matrix cores were consulted above)
    }
    std::vector<uint64_t> keys((size_t)qb * k);
    FIR_HIP(hipMemcpyAsync(keys.data(), g->dkeys, keys.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, g->stream));
    FIR_HIP(hipStreamSynchronize(g->stream));
    return fir_keys_unpack(keys.data(), qb * k, idx, dist);
}

int fir_range_distances_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                            float* d_out, void* stream) {
    if (!g || !d_out || (qb > 0 && !d_queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    return range_dev(g, d_queries, qb, start_pos, end_pos, d_out, stream ? (hipStream_t)stream : g->stream);
}

int fir_range_distances(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, float* out) {
    if (!g || !out || (qb > 0 && !queries)) return fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fail(FIR_ERR_ARG, "qb < 0");
    if (qb == 0 || g->n == 0) return FIR_OK;
    int rc = check_range(g, start_pos, end_pos);
    if (rc) return rc;
    FIR_HIP(hipSetDevice(g->device));
    if ((rc = grow(g->dq, g->dq_cap, (size_t)qb * g->d))) return rc;
    if ((rc = grow(g->dout, g->dout_cap, (size_t)qb * g->n))) return rc;
    FIR_HIP(hipMemcpyAsync(g->dq, queries, (size_t)qb * g->d * sizeof(float), hipMemcpyHostToDevice, g->stream));
    if ((rc = range_dev(g, g->dq, qb, start_pos, end_pos, g->dout, g->stream))) return rc;
    FIR_HIP(hipMemcpyAsync(out, g->dout, (size_t)qb * g->n * sizeof(float), hipMemcpyDeviceToHost, g->stream));
    FIR_HIP(hipStreamSynchronize(g->stream));
    return FIR_OK;
}

int fir_profile_enable(fir_gallery* g, int32_t on) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    g->profiling = on != 0;
    g->ev_used = 0;
    return FIR_OK;
}

int fir_profile_read(fir_gallery* g, float* ms, int32_t cap, int32_t* count, double* bytes_per_launch) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    FIR_HIP(hipSetDevice(g->device));
    const int32_t launches = (int32_t)(g->ev_used / 2);
    for (int32_t i = 0; i < launches; ++i) {
        FIR_HIP(hipEventSynchronize(g->ev[2 * i + 1]));
        float t = 0.f;
        FIR_HIP(hipEventElapsedTime(&t, g->ev[2 * i], g->ev[2 * i + 1]));
        if (ms && i < cap) ms[i] = t;
    }
    if (count) *count = launches;
    if (bytes_per_launch) *bytes_per_launch = g->last_bytes;
    g->ev_used = 0;
    return FIR_OK;
}

int fir_gallery_sync(fir_gallery* g) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    FIR_HIP(hipSetDevice(g->device));
    FIR_HIP(hipStreamSynchronize(g->stream));
    return FIR_OK;
}

int fir_gallery_value_range(fir_gallery* g, int32_t* gallery_plain, int32_t* last_queries_plain) {
    if (!g || !gallery_plain || !last_queries_plain) return fail(FIR_ERR_ARG, "fir_gallery_value_range: null argument");
    FIR_HIP(hipSetDevice(g->device));
    int32_t h[2] = {0, 0};
    FIR_HIP(hipStreamSynchronize(g->stream));
    FIR_HIP(hipMemcpy(h, g->range, sizeof h, hipMemcpyDeviceToHost));
    *gallery_plain = h[0] == 0;
    *last_queries_plain = g->q_serial != 0 && h[1] != g->q_serial;
    return FIR_OK;
}

int fir_gallery_set_tuning(fir_gallery* g, int32_t queries_per_pass, int32_t waves) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (queries_per_pass < 0) g->qpp = 0;   // back to automatic
    if (queries_per_pass > 0) {
        if (queries_per_pass != 1 && queries_per_pass != 2 && queries_per_pass != 4 && queries_per_pass != 8 &&
            queries_per_pass != 16)
            return fail(FIR_ERR_ARG, "queries_per_pass must be 1, 2, 4, 8 or 16");
        g->qpp = queries_per_pass;
    }
    if (waves < 0 || (waves % 4) != 0) return fail(FIR_ERR_ARG, "waves must be a non-negative multiple of 4");
    g->waves_req = waves;
    return FIR_OK;
}

int fir_gallery_get_tuning(const fir_gallery* g, int32_t* queries_per_pass, int32_t* waves, int32_t* max_waves) {
    if (!g) return fail(FIR_ERR_ARG, "gallery is NULL");
    if (queries_per_pass) *queries_per_pass = effective_qpp(g);
    if (waves) *waves = g->last_waves;   // waves of the most recent scan launch (0 before the first)
    if (max_waves) *max_waves = g->max_waves;
    return FIR_OK;
}

}  // extern "C"
