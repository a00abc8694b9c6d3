// fir_gemm_f16x.h -- the LDS-tile pass of the fp16 matrix-core path on v_mfma_f32_16x16x32_f16 (included by fir_gemm.hip,
// inside its anonymous namespace). Same per-wave output tile as k_gemm_proxy_f16 (32 rows x 128 queries), same bytes per
// MFMA cycle from LDS and from the gallery stream; what differs is the MFMA shape, which the chip clocks differently under
// load (MI355X_MICROARCH.md, DVFS give-back item 7: build both at the same output tile per wave, keep the faster by wall
// on random data). fir_gemm::mfma16 selects it; the fragment order of both operands is then the "16-row" one:
//
//   gallery  gh[(rb * dk16 + 2 * kk + s) * 64 + l]   row 32 rb + 16 s + (l & 15),   features 32 kk + 8 (l >> 4) + 0..7
//   queries  qh[((pair * 8 + jb) * dk32 + kk) * 64 + l]   query 128 pair + 16 jb + (l & 15),   the same features
//
// (dk32 = dk16 / 2; a row block is still dk16 consecutive one-KiB pieces, so the gallery stream and its double buffer are
// addressed exactly as in k_gemm_proxy_f16.) Accumulator tile (s, jb): f32x4, lane l holds query 16 jb + (l & 15) against
// rows 16 s + 4 (l >> 4) + 0..3 of the block.
#pragma once
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// rows of the tiled f32 gallery * scale -> 16-row fragment order (see above); kmax as k_gemm_pack_gallery_f16
__global__ void __launch_bounds__(256) k_gemm_pack_gallery_f16x(const float4* __restrict__ gal4, int64_t n, int dp4, int dk16, float scale,
                                                                 uint4* __restrict__ gh, int kmax) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (rb, piece, lane)
    const int64_t rblocks = (n + 31) / 32;
    if (o >= rblocks * dk16 * 64) return;
    const int l = (int)(o & 63);
    const int64_t t = o >> 6;
    const int piece = (int)(t % dk16);
    const int64_t rb = t / dk16;
    const int kk = piece >> 1, s = piece & 1;
    const int64_t row = rb * 32 + 16 * s + (l & 15);
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * kk + 8 * (l >> 4) + j;
        float x = 0.f;
        if (row < n && k < kmax) {
            const float4 g = gal4[((row >> 6) * dp4 + (k >> 2)) * 64 + (row & 63)];
            x = (k & 3) == 0 ? g.x : (k & 3) == 1 ? g.y : (k & 3) == 2 ? g.z : g.w;
        }
        v[j] = (_Float16)(x * scale);
    }
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    gh[o] = u;
}

// queries * qmul -> 16-row fragment order; blockIdx.y = pair of 64-query passes
// qmap (a second-chance round, fir_gemm_fb.h): slot i of the one pair is query qmap[i] of the call, state[0] - live_off slots are live
__global__ void __launch_bounds__(256) k_gemm_pack_queries_f16x(const float* q, int nq, int d, int dk16, const float* __restrict__ qmul, uint4* qh,
                                                                 int qstride, const int* __restrict__ qmap = nullptr, const int* __restrict__ state = nullptr,
                                                                 int live_off = 0) {
    if (qmap) {
        nq = state[0] - live_off;
        if (nq <= 0) return;
        nq = nq < 2 * kQT ? nq : 2 * kQT;
    }
    const int dk32 = dk16 >> 1;
    const int q_base = (int)blockIdx.y * 2 * kQT;
    qh += (size_t)blockIdx.y * 8 * dk32 * 64;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 8 * dk32 * 64) return;
    const int l = o & 63;
    const int t = o >> 6;
    const int kk = t % dk32, jb = t / dk32;
    const int qi = q_base + jb * 16 + (l & 15);
    const float mul = qi < nq ? qmul[qi] : 0.f;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * kk + 8 * (l >> 4) + j;
        const float x = (qi < nq && k < d) ? q[(size_t)(qmap ? qmap[qi] : qi) * qstride + k] : 0.f;
        v[j] = (_Float16)(x * mul);
    }
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    qh[o] = u;
}

// One wave: 32 rows x 128 queries. Parameters, grid shapes and the `share` placement as k_gemm_proxy_f16.
// ODD = units (of kRing pieces) per row block is odd: only then do the two gallery buffers end a row block in swapped roles
// and need a copy; with the even form compiled separately the common row lengths (256, 512, 1280 features) carry neither the
// copy nor the merge point the compiler hung an s_waitcnt vmcnt(0) on.
// MODE 0: one minimum per (row block, query) of rows [row_begin, row_end) -> sample (the order-statistic flow, k_gemm_tau; not
// instantiated any more: the 16-row kernels always run the smallest-proxy flow);
// MODE 3: the full pass with the threshold found ON THE WAY (top-1; no sample pass, no threshold kernel): per query the ranks
// keep T = (smallest proxy seen so far + |q|^2) + one rounding window, as float bits >= 0 so that an unsigned atomicMin orders
// them -- in LDS per workgroup, mirrored to `smin` (global) by atomicMin whenever a workgroup lowers it (that is where the final
// bound comes from); the workgroups exchange once, behind their warm-up walk, which leaves every one of them with the minimum
// over the first row blocks of about half the chip (tens of thousands of rows: what the sample pass used to provide) -- a
// re-read of `smin` into a register held across the row block measured slower than the appends it saved (256 VGPRs: it spilled a
// loop-carried pointer), and loads of `smin` -- plain, sc1 or LDS-DMA -- turned out to be served from stale lines of the XCD's own
// L2; each wave therefore posts-and-fetches its sixteen queries' T with ONE returning atomicMin on a thinning schedule
// (profiles/r03_adaptive_threshold.txt). A row is appended when its proxy is below tq = T - |q|^2 AT THAT MOMENT. T only ever
// falls, so whatever was not appended has a proxy >= fl(T_final - |q|^2) =: tau, which k_gemm_adapt_final hands to the
// re-rank's certificate; and every row within the window of the smallest proxy of ALL rows IS appended (its proxy is below
// every T the pass ever held). With T = +inf at the start the first row block of a wave would append all its rows: its sums
// are therefore looked at twice -- once only lowering T (then the workgroup exchanges T with `smin`), then, still in the
// registers, for real. Here `tau` carries the windows, `sample` the |q|^2, `smin` the global T.
// MODE 1: the full pass, every row below tau is appended; MODE 2: the sample of the smallest-proxy flow -- row blocks
// 0, rb_stride, 2 rb_stride, ... of the gallery ((row_end - row_begin) / 32 of them, spread over all of it: the reference's
// galleries are ordered by class), smin[q] <- the smallest proxy seen (fir::f32_orderable bits, atomicMin, caller presets +inf);
// with sub_stride != 0 (top-K) as kRtSubsets disjoint subsets' minima, smin[subset * sub_stride + q] (k_gemm_tau_kmin).
// A unit is kRing = 8 gallery pieces = four 32-feature steps; per step two A fragments (the row halves) against eight B
// fragments: sixteen MFMAs of 16 cycles. The eight B fragments are ONE register set that rolls: fragment j is re-read for
// the next step right behind the two MFMAs that used it, fourteen MFMAs before its next use.
// Bit 1 of `nt_flags` (experiment, FIR_GEMM_STAGGER): waves 4-7 -- the partners of waves 0-3 on their SIMDs -- run half a unit of
// throw-away MFMAs first, so that partners do not reach their epilogues and their end-of-unit waits together. Bit 0: the
// gallery stream is read once per launch (non-temporal loads).
constexpr int kXStage = 16;             // staged appends per query and workgroup
// NJB: query blocks of 16 the wave multiplies against (8 = the whole 128-query tile). A call of at most 16 / 32 queries fills one / two
// blocks; with NJB = 1 / 2 the pass does an eighth / a quarter of the MFMAs and LDS reads and is what such a call should be: one read
// of the fp16 fragments at the rate the memory system gives (the tile's layout in LDS and in `qh` is unchanged).
// DBG (timing experiments only, FIR_GEMM_DBG_SKIP; own instantiations so that the production kernels' register allocation is not
// touched -- as runtime flags the two tests made the row loop spill): bit 0 = no epilogue, bit 1 = no gallery stream, bit 2 = no
// re-read of the query fragments, bit 5 = no MFMAs. The answers of such a kernel are wrong.
template <int MODE, int STREAMED, int ODD, int DBG = 0, int NJB = 8>
__global__ void __launch_bounds__(kGemmBlock, 1) k_gemm_proxy_f16x(const uint4* __restrict__ gh, const float* __restrict__ gnorm, const uint4* qh,
                                                                    const float* __restrict__ qinv, int64_t n, int64_t row_begin, int64_t row_end,
                                                                    int dk16, const float* tau, unsigned long long* lists, int* counts,
                                                                    float* sample, int sample_rows, int share, int nt_flags,
                                                                    int rb_stride, unsigned int* smin, int sub_stride) {
    extern __shared__ __attribute__((aligned(16))) uint4 lqx[];
    // MODE 1 as a second-chance round (fir_gemm_fb.h): `smin` is the list's length, `sub_stride` the part of it earlier rounds took --
    // nothing left for this round: every workgroup returns at once (a kernel boundary lies between the writers and this load)
    if (MODE == 1 && smin != nullptr && (int)smin[0] - sub_stride <= 0) return;
    const int nt = nt_flags & 1;
    __shared__ float tau_s[2 * kQT], qinv_s[2 * kQT];
    // MODE 1: appends are staged per workgroup in LDS (an LDS atomic counts in lgkmcnt and returns in ~100 cycles; a returning GLOBAL
    // atomic sits in the in-order vmcnt queue behind the prefetched gallery loads -- every append drained the wave's stream) and
    // flushed to the global lists once, at the end; a query that fills its kXStage slots appends directly (rare, correct)
    // MODE 4: MODE 3 for the K nearest rows (K = sub_stride, 2..8). Per query EIGHT slot minima over disjoint row sets -- slot i = the rows a
    // lane holds in position i of its eight (rows 16 (i >> 2) + 4 (lane >> 4) + (i & 3) of every row block): four rows of EVERY block feed
    // every slot, so a cluster of a few dozen near rows fills all eight at once (slots by row block left the slots of a class-ordered
    // training set to different classes: profiles/r04_knn_matrix_cores.txt) -- the K smallest slot minima belong to K distinct rows, so T = (the K-th smallest slot minimum + |q|^2) +
    // window bounds the K-th smallest proxy of all rows from above -- what the K-nearest re-rank's certificate needs (k_gemm_rerank_topk:
    // window hung on the K-th smallest proxy of the list). Every slot only falls, so T only falls, and the MODE 3 argument carries over
    // word for word. The slots live in LDS (slot_s, as the float bits of (slot minimum + |q|^2) + window, >= 0) and in `smin` (8 words
    // per query), exchanged by the same atomics on the same schedule; the hot path still compares against one bound per query (tq_s / tqr).
    constexpr bool kAdapt = MODE == 3 || MODE == 4;
    constexpr bool kSlots = MODE == 4;
    const int kth = kSlots ? sub_stride : 1;
    // the K-th smallest of a query's eight slot words (rank by value, then slot: every word gets a distinct rank)
    auto kth_of_slots = [&](const unsigned int* sl) -> unsigned int {
        const uint4 lo = *(const uint4*)sl, hi = *(const uint4*)(sl + 4);          // (sl is 32-byte aligned: a query's eight words)
        const unsigned int v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        unsigned int res = 0xFFFFFFFFu;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) rank += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
            res = rank == kth - 1 ? v[i] : res;
        }
        return res;
    };
    constexpr bool kAppend = MODE == 1 || kAdapt;
    __shared__ unsigned long long skeys[kAppend ? 2 * kQT * kXStage : 1];
    __shared__ int scnt[kAppend ? 2 * kQT : 1];
    __shared__ float qn_s[kAdapt ? 2 * kQT : 1], win_s[kAdapt ? 2 * kQT : 1];
    __shared__ __attribute__((aligned(16))) unsigned int slot_s[kSlots ? 2 * kQT * 8 : 8];
    // MODE 3: tq_s = T - |q|^2 as the hot path compares it, rewritten by whoever lowers T (tau_s holds T itself). A racing store may leave
    // the value of an OLDER (larger) T: harmless -- any T the pass ever held is >= the final one, so whatever fails the test against
    // it has a proxy >= fl(T_final - |q|^2), the bound the certificate is given
    __shared__ float tq_s[kAdapt ? 2 * kQT : 1];
    int pair_of_wg = (int)blockIdx.y;
    int64_t rg_first = blockIdx.x, rg_step = gridDim.x, rg_last = -1;
    int range = (int)blockIdx.x;
    if (share > 0) {
        const int w = (int)blockIdx.x, xcd = w & 7, slot = w >> 3;
        const int ranges = ((int)gridDim.x >> 3) / share * 8;
        range = xcd + 8 * (slot / share);
        if (range >= ranges) return;                                       // uniform per workgroup
        pair_of_wg = slot % share;
        const int64_t nrg_all = (((row_end + 31) / 32 - row_begin / 32) + (blockDim.x >> 6) - 1) / (blockDim.x >> 6);
        rg_first = nrg_all * range / ranges;
        rg_last = nrg_all * (range + 1) / ranges;
        rg_step = 1;
    }
    const int dk32 = dk16 >> 1;
    {
        const size_t pr = (size_t)pair_of_wg;
        qh += pr * 8 * dk32 * 64;
        qinv += pr * 2 * kQT;
        tau += pr * 2 * kQT;
        lists += pr * 2 * kQT * kListCap;
        counts += pr * 2 * kQT;
        if (MODE == 0) sample += pr * 2 * kQT * ((sample_rows + 31) / 32);
        if (MODE == 2 || MODE == 3) smin += pr * 2 * kQT;
        if (MODE == 4) smin += pr * 2 * kQT * 8;
        if (kAdapt) sample += pr * 2 * kQT;
    }
    const int64_t rbs = MODE == 2 ? rb_stride : 1;                        // gallery row blocks per row block of the pass
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wpb = blockDim.x >> 6;
    if (threadIdx.x < 2 * kQT) {
        if (kSlots) {
            unsigned int v8[8];
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) v8[sl] = atomicMin(&smin[threadIdx.x * 8 + sl], 0xFFFFFFFFu);
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) slot_s[threadIdx.x * 8 + sl] = v8[sl];
            tau_s[threadIdx.x] = __uint_as_float(kth_of_slots(&slot_s[threadIdx.x * 8]));
        } else
        tau_s[threadIdx.x] = MODE == 1 ? tau[threadIdx.x] : MODE == 3 ? __uint_as_float(atomicMin(&smin[threadIdx.x], 0xFFFFFFFFu)) : 0.f;    // (MODE 3: an atomic, see the refresh below)
        qinv_s[threadIdx.x] = qinv[threadIdx.x];
        if (kAppend) scnt[threadIdx.x] = 0;
        if (kAdapt) {
            qn_s[threadIdx.x] = sample[threadIdx.x];
            win_s[threadIdx.x] = tau[threadIdx.x];
            tq_s[threadIdx.x] = tau_s[threadIdx.x] - qn_s[threadIdx.x];
        }
    }
    const int64_t rb_begin = row_begin / 32, rb_end = (row_end + 31) / 32;
    const int64_t nrg = (rb_end - rb_begin + wpb - 1) / wpb;
    // a unit of the few-block forms is SIXTEEN gallery pieces (eight steps): such a pass is one read of the fragments and nothing else, and
    // what it reads at is set by the bytes a wave keeps in flight -- 16 KiB instead of 8 (registers are not scarce with one or two
    // query blocks). Resident tiles only (the streamed ring's slots and its vmcnt(12) are sized for eight pieces).
    constexpr int RING = (!STREAMED && NJB < 8) ? 2 * kRing : kRing;
    const int units = dk16 / RING;
    const int64_t rg_end = rg_last >= 0 ? rg_last : nrg;
    int64_t rg = rg_first;
    if (rg >= rg_end) return;                        // uniform per workgroup
#define FIR_X_LD(P) (nt ? ld_nt(P) : *(P))
    // (wave-uniform block pointers: the lane index is added in the load itself, so that the loads take the scalar-base form --
    // one 32-bit lane offset register instead of a 64-bit address per pointer)
#define FIR_X_BLOCK(RG) (gh + (size_t)(((rb_begin + (RG) * wpb + wave) < rb_end ? (rb_begin + (RG) * wpb + wave) : rb_end - 1) * rbs) * dk16 * 64)
    const uint4* a_cur = FIR_X_BLOCK(rg);
    uint4 cur[RING], nxt[RING];
#pragma unroll
    for (int u = 0; u < RING; ++u) cur[u] = FIR_X_LD(a_cur + (size_t)u * 64 + lane);
    constexpr int kUnitsPerSlab = kSlabH / RING;                         // units of the LDS-resident tile (512 features)
    const bool resident = !STREAMED;                                      // (the caller streams whatever does not fit: dk16 > kSlabH)
    // STREAMED (rows longer than the 512 features whose 128-query tile fits LDS): the query fragments go through a ring of FOUR
    // 32-KiB LDS slots of one unit (four steps x eight query blocks) each. Units are numbered c = 0, 1, 2, ... over the whole walk of
    // the workgroup (slab of unit c: c mod units); slot c & 3 holds unit c. The slab of unit c + 3 is requested DURING unit c -- this
    // wave's four one-KiB pieces by LDS-DMA, one behind each step's MFMAs (a request holds the issuing wave for 60-180 cycles: spread
    // out, the SIMD's other wave covers them) -- into the slot unit c - 1 was read from, which every wave left before the barrier
    // that ended unit c - 1. At the end of unit c every wave waits for its pieces of unit c + 2 (requested a whole unit earlier:
    // vmcnt(12) = this unit's eight gallery loads and four requests may stay in flight) and ONE barrier publishes them; so the
    // fragments of unit c + 1's first step, which the rolling re-read fetches during unit c's last step, were published one barrier
    // earlier, and the fragment pipeline runs on through unit borders as in the resident form -- the barrier no longer drains it.
    // The requests are inline assembly: the compiler orders every LDS read behind a pending LDS-DMA it knows about with
    // s_waitcnt vmcnt(0), and it cannot see that ring slots do not alias. Its own counted waits for the gallery loads stay
    // correct with requests it does not know in the queue: vmcnt(N) leaves the N YOUNGEST operations outstanding, whatever they are.
    const uint32_t lds_base = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)lqx;
    auto request_piece = [&](int hq, int slot, int i) {              // piece i of this wave: (step i, query block `wave`) of slab hq
        const uint4* src = qh + ((size_t)wave * dk32 + (size_t)hq * 4 + i) * 64 + lane;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + (uint32_t)slot * (4 * RING * 1024) + (uint32_t)(i * 8 + wave) * 1024);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
    };
    int ring_c = 0, hq3 = 0;                         // unit counter of the walk; slab index of unit ring_c + 3
    if (STREAMED) {
        int hq = 0;
        for (int cc = 0; cc < 3; ++cc) {             // units 0, 1, 2 before the walk starts
#pragma unroll
            for (int i = 0; i < 4; ++i) request_piece(hq, cc, i);
            hq = hq + 1 == units ? 0 : hq + 1;
        }
        hq3 = hq;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // (also the prologue's gallery loads: once per kernel)
        __syncthreads();
    } else {
        // LDS image: step-major, the eight query blocks of a step side by side -- (kk * 8 + jb) * 1 KiB
        for (int i = threadIdx.x; i < NJB * dk32 * 64; i += blockDim.x) {                   // (the live query blocks: NJB of the tile's eight)
            const int jb = i / (dk32 * 64), r = i - jb * dk32 * 64;
            lqx[(size_t)((r >> 6) * 8 + jb) * 64 + (r & 63)] = qh[(size_t)jb * dk32 * 64 + r];
        }
        __syncthreads();
    }
    float smallest[NJB];                             // MODE 2: running minima of this lane's NJB queries
#pragma unroll
    for (int j = 0; j < NJB; ++j) smallest[j] = __builtin_huge_valf();
    uint4 B[NJB];                                    // the fragments of the first step of the first unit (slot 0 / the resident tile's start)
#pragma unroll
    for (int j = 0; j < NJB; ++j) B[j] = lqx[lane + j * 64];
    if (NJB == 8 && (nt_flags & 2) && resident && wave >= wpb / 2) {
        // half a unit of MFMAs whose result goes nowhere the kernel's outputs are computed from: it only delays this wave
        f32x4 junk = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 32; ++r) junk = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(B[r & (NJB - 1)]), as_f16x8(B[(r + 1) & (NJB - 1)]), junk, 0, 0, 0);
        if (nt_flags & 32)                               // (FIR_GEMM_STAGGER=2: half a row block at 512 features instead of half a unit)
            for (int r = 0; r < 96; ++r) junk = __builtin_amdgcn_mfma_f32_16x16x32_f16(as_f16x8(B[r & (NJB - 1)]), as_f16x8(B[(r + 1) & (NJB - 1)]), junk, 0, 0, 0);
        asm volatile("" ::"v"(junk));
    }
    // The append forms DEFER a full row block's epilogue into the next row block's first step: the first step's MFMAs start the sums
    // from a zero C operand, so query block j's sixteen sums stay readable until the two MFMAs of index j of that step overwrite
    // them -- the check of query block j (four v_max3, an fma, a compare; rarely the eight proxies and an append) sits right in
    // front of them and runs beside the MFMAs of the blocks before it, instead of leaving the matrix pipe to a partner wave that,
    // alone, waits for its own LDS reads (profiles/r03_gemm_time_decomposition.txt: the epilogue cost 0.18 of 1.56 ms). DBG & 8: the
    // epilogue where it was, for A/B runs.
    constexpr bool kDefer = kAppend && !(DBG & 8) && !(DBG & 1);
    f32x4 acc[2][NJB];                               // (first written by the MFMAs of a row block's first step, against a zero C operand)
    bool pend = false;                               // a full row block's checks are still owed
    int64_t p_rb = 0;
    float4 pg[2] = {};
    float p_gmin = 0.f;
    float m2r[kDefer ? NJB : 1], tqr[kDefer ? NJB : 1];  // per query block of this lane: 2 / scale, and the bound as of the last row block's end
    if (kDefer) {
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            m2r[jb] = 2.0f * qinv_s[jb * 16 + (lane & 15)];
            tqr[jb] = kAdapt ? tq_s[jb * 16 + (lane & 15)] : tau_s[jb * 16 + (lane & 15)];
        }
    }
    bool warm = kAdapt;                              // MODE 3 / 4: the first row block's sums are looked at twice (below)
    // MODE 4: the K-th smallest slot value as the new T of query q (a value read a moment ago is >= the slot's current one, so this is a bound)
    auto slots_to_T = [&](int q) {
        const unsigned int tk = kth_of_slots(&slot_s[q * 8]);
        if (tk < atomicMin((unsigned int*)&tau_s[q], tk)) tq_s[q] = __uint_as_float(tk) - qn_s[q];
    };
    // a new smallest proxy `mn` (= the smallest of the lane's eight proxies pv) for query q: T falls, here and (through `smin`) for everybody else
    auto lower_T = [&](int q, const float (&pv)[8], float mn) {
        if (!kSlots) {
            const float tn = fmaxf((mn + qn_s[q]) + win_s[q], 0.f);
            if (tn < tau_s[q]) {
                if (__float_as_uint(tn) < atomicMin((unsigned int*)&tau_s[q], __float_as_uint(tn))) tq_s[q] = tn - qn_s[q];
                atomicMin(&smin[q], __float_as_uint(tn));
            }
        } else {
            // (the eight slot words in ONE round trip to the LDS, as two 16-byte reads: read one by one between the conditional atomics
            // they cost an append event eight round trips in a row. A slot another wave lowers in between is compared against its
            // older, larger value: one atomic more, the same minimum.)
            const uint4 s_lo = *(const uint4*)&slot_s[q * 8], s_hi = *(const uint4*)&slot_s[q * 8 + 4];
            const unsigned int have[8] = {s_lo.x, s_lo.y, s_lo.z, s_lo.w, s_hi.x, s_hi.y, s_hi.z, s_hi.w};
            const float qn = qn_s[q], win = win_s[q];
            bool fell = false;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float tn = fmaxf((pv[i] + qn) + win, 0.f);        // (a NaN proxy: fmaxf(NaN, 0) = 0 -- kept out by the test on pv[i] itself)
                if (pv[i] == pv[i] && tn < __uint_as_float(have[i])) {
                    atomicMin(&slot_s[q * 8 + i], __float_as_uint(tn));
                    atomicMin(&smin[q * 8 + i], __float_as_uint(tn));
                    fell = true;
                }
            }
            if (fell) slots_to_T(q);
        }
    };
    // the workgroup's first-row-block minima go out, everybody else's come in (uniform over the workgroup: every wave's first row block
    // is its warm-up block)
    auto exchange_T = [&]() {
            // the workgroup's warm-up minima go out, everybody else's come in
        __syncthreads();
        if (threadIdx.x < 2 * kQT) {
            if (kSlots) {
                unsigned int mine[8], old[8];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) mine[sl] = slot_s[threadIdx.x * 8 + sl];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) old[sl] = (nt_flags & 8) ? mine[sl] : atomicMin(&smin[threadIdx.x * 8 + sl], mine[sl]);
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) slot_s[threadIdx.x * 8 + sl] = old[sl] < mine[sl] ? old[sl] : mine[sl];
                tau_s[threadIdx.x] = __uint_as_float(kth_of_slots(&slot_s[threadIdx.x * 8]));
            } else {
            const unsigned int mine = __float_as_uint(tau_s[threadIdx.x]);
            const unsigned int old = (nt_flags & 8) ? mine : atomicMin(&smin[threadIdx.x], mine);
            tau_s[threadIdx.x] = __uint_as_float(old < mine ? old : mine);
            }
            tq_s[threadIdx.x] = tau_s[threadIdx.x] - qn_s[threadIdx.x];
        }
        __syncthreads();
    };
    int blk_no = 0;
    int dbg_units = 0;                               // (DBG & 256: units walked so far)
    int touch_e = 0, touch_o = 0;                    // (DBG & 512)
    int64_t rg_next = rg;
    for (; rg < rg_end; rg = rg_next) {
        const bool warm_it = warm;                        // this wave's first row block: T is still what it was preset to
        warm = false;
        rg_next = rg + rg_step;
        const int64_t rbp = rb_begin + rg * wpb + wave;   // row block of the pass ...
        const int64_t rb = rbp * rbs;                     // ... and of the gallery
        const bool active = rbp < rb_end;
        const int64_t rgn = rg_next;
        const uint4* a_nxt = FIR_X_BLOCK(rgn < rg_end ? rgn : rg);
        if (kAdapt) {
            if (!warm_it && wave < NJB && !(nt_flags & 4)) {
                // What the other workgroups have reached since. Only ATOMICS read `smin`: they execute at the memory side, so what they
                // return is the value every XCD's updates have been folded into -- a load, even an sc1 one, can be served by a line this
                // XCD's L2 took in earlier (measured: whole XCDs' workgroups never saw the others' bounds and appended 16 rows each per
                // query, 4 000+ per query with one pair over 256 row ranges). One returning atomicMin posts this wave's sixteen
                // queries' T and fetches the global one; its result is used at once (nothing is held across the row block -- at 256
                // registers that spills), on a schedule that thins out: row blocks 2, 3, 4, 6, 8, 12, 16, 24, ... (T settles early).
                const int v = blk_no >> __builtin_ctz((unsigned)blk_no | 0x40000000u);
                if ((v == 1 || v == 3) && lane < 16) {
                    int zl;                                   // (an opaque zero: keeps the addresses from being hoisted out of the row loop and spilled)
                    asm volatile("v_mov_b32 %0, 0" : "=v"(zl));
                    const int ql = 16 * wave + lane + zl;
                    if (kSlots) {
                        // (all eight round trips in flight together: one after the other they cost a refresh eight memory latencies)
                        unsigned int mine[8], old[8];
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl) mine[sl] = slot_s[ql * 8 + sl];
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl) old[sl] = atomicMin(&smin[ql * 8 + sl], mine[sl]);
                        bool fell = false;
#pragma unroll
                        for (int sl = 0; sl < 8; ++sl)
                            if (old[sl] < mine[sl]) { atomicMin(&slot_s[ql * 8 + sl], old[sl]); fell = true; }
                        if (fell) slots_to_T(ql);
                    } else {
                    const unsigned int mine = __float_as_uint(tau_s[ql]);
                    const unsigned int old = atomicMin(&smin[ql], mine);
                    if (old < mine && old < atomicMin((unsigned int*)&tau_s[ql], old)) tq_s[ql] = __uint_as_float(old) - qn_s[ql];
                    }
                }
            }
            ++blk_no;
        }
        float4 gns[2];                               // squared norms of rows 16 s + 4 (lane >> 4) + 0..3 of the block
        const bool full_block = active && rbp * 32 >= row_begin && rbp * 32 + 32 <= row_end && rb * 32 + 32 <= n && (MODE != 0 || rb * 32 + 32 <= sample_rows);
        // MODE 2, sub_stride: eight positions x eight waves = kRtSubsets disjoint subsets of the sampled rows (below)
        unsigned int* smin_blk = MODE == 2 ? smin + (size_t)(wave & 7) * sub_stride : nullptr;       // (position 0's subset of this wave)
        // one query block of a full row block `rbq` against the bound (the append forms, not the warm-up walk); g0 / g1 / gminq = the
        // block's row norms and their minimum
        auto check_jb = [&](int jb, int64_t rbq, const float4 g0, const float4 g1, float gminq) {
            const int q = jb * 16 + (lane & 15);
            // (kDefer: the lane's scale and bound come from registers -- an LDS read here would wait, in order, behind the query
            // fragments just requested for the next step; a bound that is a row block old is a larger one: it appends more, never less)
            const float m2 = kDefer ? m2r[jb] : 2.0f * qinv_s[q];
            const float tq = kDefer ? tqr[jb] : kAdapt ? tq_s[q] : tau_s[q];
            if (!(nt_flags & 64)) {
                // with m2 > 0, fl(gmin - m2 amax) <= fl(gn_r - m2 acc_r) for every row r of the lane (gmin <= gn_r, amax >= acc_r, rounding is
                // monotone): `lb >= tq` proves that no row of the block is appended -- seven instructions instead of sixteen. NaN sums never
                // enter amax, as they never enter the minimum below. (v_max3_f32 by hand: fmaxf() on MFMA results quiets every operand first.)
                float amax;
                asm("v_max3_f32 %0, %1, %2, %3\n\tv_max3_f32 %0, %0, %4, %5\n\tv_max3_f32 %0, %0, %6, %7\n\tv_max_f32 %0, %0, %8"
                    : "=&v"(amax)
                    : "v"(acc[0][jb][0]), "v"(acc[0][jb][1]), "v"(acc[0][jb][2]), "v"(acc[0][jb][3]), "v"(acc[1][jb][0]), "v"(acc[1][jb][1]),
                      "v"(acc[1][jb][2]), "v"(acc[1][jb][3]));
                if (!(__builtin_fmaf(-m2, amax, gminq) < tq)) return;
            }
            const float gnv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            float pv[8];
            float mn = __builtin_huge_valf();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pv[i] = __builtin_fmaf(-m2, acc[i >> 2][jb][i & 3], gnv[i]);
                mn = fminf(mn, pv[i]);                                   // NaN never enters, like k_gemm_tau's ordering
            }
            if (mn < tq) {
                // The rare path works on its OWN copy of the query index, made here from the hardware's lane count: every address below
                // (scnt, skeys, the slots, the lists) is then computed where it is used. From `q` itself the compiler hoisted eight sets
                // of them out of the row loop, spilled them, and reloaded them here -- a scratch load and an s_waitcnt vmcnt(0) per
                // appended row, i.e. every append waited for the gallery prefetch in flight (~1.2 us per event:
                // profiles/r04_append_path.txt).
                int lz;                                               // the lane index, from the hardware: no register of the hot loop is read
                asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lz));
                const int qr = jb * 16 + (lz & 15);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (pv[i] < tq) {
                        const int64_t row = rbq * 32 + 16 * (i >> 2) + 4 * (lz >> 4) + (i & 3);
                        const unsigned long long key = fir::key_pack(pv[i], (uint32_t)row);
                        const int st = atomicAdd(&scnt[qr], 1);
                        if (st < kXStage) {
                            skeys[qr * kXStage + st] = key;
                        } else {
                            const int slot = atomicAdd(&counts[qr], 1);
                            if (slot < kListCap) lists[(size_t)qr * kListCap + slot] = key;
                            // the list is full: this query is uncertified whatever else happens (fir_gemm_fb.h gives it a second pass), so the
                            // workgroup stops appending for it -- a pass whose bound stays loose must not turn into millions of atomics on
                            // one counter (~11 ns each: a whole gallery's worth is 11 ms). The next lowering of T writes tq_s again.
                            else if (kAdapt) tq_s[qr] = -__builtin_huge_valf();
                        }
                    }
                }
                if (kAdapt) lower_T(qr, pv, mn);
            }
        };
        auto unit = [&](uint4 (&C)[RING], uint4 (&N)[RING], int h, auto first_tag) {
            constexpr bool kFirst = decltype(first_tag)::value;      // the first unit of the row block: its first step starts the sums
            const uint4* src = h + 1 < units ? a_cur + (size_t)(h + 1) * RING * 64 : a_nxt;
            const uint4* bq = STREAMED ? lqx + lane + (size_t)(ring_c & 3) * 4 * RING * 64 : lqx + lane + (size_t)(h % kUnitsPerSlab) * RING * 4 * 64;
            // The next unit's eight gallery pieces are requested in one burst in front of the unit. (DBG & 16, measured and not kept: two per
            // step, behind the fourth and the eighth pair of MFMAs -- every piece still exactly one unit before its use -- ran 6 % slower at
            // 512 features and 5 % at 256: profiles/r03_gemm_time_decomposition.txt.)
            unsigned long long t_burst0 = 0;
            if (DBG & 256) t_burst0 = __builtin_amdgcn_s_memtime();
            if (!(DBG & 16) && !(DBG & 2) && !(DBG & 64)) {
                if (nt) {
#pragma unroll
                    for (int u = 0; u < RING; ++u) N[u] = ld_nt(src + (size_t)u * 64 + lane);
                } else {
#pragma unroll
                    for (int u = 0; u < RING; ++u) N[u] = src[(size_t)u * 64 + lane];
                }
            }
            if ((DBG & 512) && pair_of_wg == 0) {
                // (experiment, audit build: ONE of the workgroups that share a row range asks for the unit after next -- one dword per 128-byte
                // line, 64 lines = the unit's 8 KiB -- so that the fragments are in the XCD's L2 when the sixteen readers' bursts come for them;
                // the value is looked at two units later, by the next unit of the same parity)
                const uint4* tgt = h + 2 < units ? a_cur + (size_t)(h + 2) * RING * 64 : a_nxt + (size_t)(h + 2 - units) * RING * 64;
                int& tch = (h & 1) ? touch_o : touch_e;
                asm volatile("" ::"v"(tch));
                tch = *(const int*)(tgt + (size_t)lane * 8);
            }
            if (DBG & 256) {
                // (timing probe, audit build: when this wave started issuing the unit's eight loads and when the last of them had been issued --
                // s_memtime is a scalar instruction: it issues in order behind them -- for the first 240 units of workgroup 0's waves, kept in the
                // unused tail of the pair's last candidate list: profiles/r04_wave_phases.txt)
                const unsigned long long t_burst1 = __builtin_amdgcn_s_memtime();
                if (blockIdx.x == 0 && dbg_units < 240 && lane == 0) {
                    unsigned long long* dst = lists + (size_t)(2 * kQT - 1) * kListCap + 2048 + (size_t)wave * 256;
                    dst[dbg_units] = (t_burst0 << 20) | ((t_burst1 - t_burst0) & 0xFFFFFull);
                }
                ++dbg_units;
            }
            if (h == units - 1 && full_block) {
                // (wave-uniform base + a lane offset made here: hoisted out of the row loop, `gnorm + 4 (lane >> 4)` is a 64-bit register
                // pair per lane that the K-nearest form spilled -- a scratch reload and an s_waitcnt vmcnt(0) in every row block)
                unsigned int zl;
                asm volatile("v_mov_b32 %0, 0" : "=v"(zl));
                const float* gb = gnorm + rb * 32;
                const unsigned int off = 4u * ((unsigned int)lane >> 4) + zl;
                gns[0] = *(const float4*)(gb + off);
                gns[1] = *(const float4*)(gb + off + 16);
            }
            const uint4* bq_after = STREAMED ? lqx + lane + (size_t)((ring_c + 1) & 3) * 4 * RING * 64
                                             : lqx + lane + (size_t)((h + 1 < units ? h + 1 : 0) % kUnitsPerSlab) * RING * 4 * 64;
#pragma unroll
            for (int t = 0; t < RING / 2; ++t) {
                const uint4* bn = t + 1 < RING / 2 ? bq + (size_t)(t + 1) * 8 * 64 : bq_after;
                const f16x8 a0 = as_f16x8(C[2 * t]), a1 = as_f16x8(C[2 * t + 1]);
#pragma unroll
                for (int j = 0; j < NJB; ++j) {
                    if (kDefer && kFirst && t == 0) {
                        if (pend) check_jb(j, p_rb, pg[0], pg[1], p_gmin);      // the previous row block's sums of query block j, about to be overwritten
                    }
                    const f16x8 b = as_f16x8(B[j]);
                    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                    if (!(DBG & 32)) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b, kFirst && t == 0 ? zero : acc[0][j], 0, 0, 0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b, kFirst && t == 0 ? zero : acc[1][j], 0, 0, 0);
                    } else {                                      // (DBG & 32: no MFMAs -- the operands are only kept alive: what the two streams cost by themselves)
                        asm volatile("" ::"v"(a0), "v"(a1), "v"(b));
                        if (kFirst && t == 0) { acc[0][j] = zero; acc[1][j] = zero; }
                    }
                    if (!(DBG & 4)) B[j] = bn[j * 64];
                    if ((DBG & 64) && t == 0 && RING == 8) {     // (DBG & 64: the next unit's pieces one at a time, behind each MFMA pair of the unit's first step)
                        N[j] = nt ? ld_nt(src + (size_t)j * 64 + lane) : src[(size_t)j * 64 + lane];
                    }
                    if ((DBG & 16) && !(DBG & 2) && (j == 3 || j == 7)) {
                        const int u = 2 * t + (j == 7 ? 1 : 0);
                        N[u] = nt ? ld_nt(src + (size_t)u * 64 + lane) : src[(size_t)u * 64 + lane];
                    }
                    __builtin_amdgcn_sched_barrier(0);            // the re-read stays right behind its fragment's last use
                }
                if (STREAMED) request_piece(hq3, (ring_c + 3) & 3, t);      // unit ring_c + 3's slab, into the slot unit ring_c - 1 has left
            }
            if (STREAMED) {
                asm volatile("s_waitcnt vmcnt(12)" ::: "memory");          // this wave's pieces of unit ring_c + 2 have landed ...
                __builtin_amdgcn_s_barrier();                               // ... and everyone's: published
                ++ring_c;
                hq3 = hq3 + 1 == units ? 0 : hq3 + 1;
            }
        };
        if (nt_flags & 16) __builtin_amdgcn_s_setprio(2);       // (experiment, FIR_GEMM_PRIO: the wave in its MFMA phase goes ahead of its partner's epilogue)
        if (!ODD) {
            unit(cur, nxt, 0, std::true_type());
            pend = false;
            unit(nxt, cur, 1, std::false_type());
            for (int h = 2; h < units; h += 2) {
                unit(cur, nxt, h, std::false_type());
                unit(nxt, cur, h + 1, std::false_type());
            }
        } else {
            unit(cur, nxt, 0, std::true_type());
            pend = false;
            for (int h = 1; h + 1 < units; h += 2) {
                unit(nxt, cur, h, std::false_type());
                unit(cur, nxt, h + 1, std::false_type());
            }
#pragma unroll
            for (int u = 0; u < RING; ++u) cur[u] = nxt[u];         // an odd number of units: the next row block's first unit sits in nxt
        }
        a_cur = a_nxt;
        if (nt_flags & 16) __builtin_amdgcn_s_setprio(0);
        if (DBG & 1) {                   // (no epilogue: the sums are only kept alive)
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) asm volatile("" ::"v"(acc[0][jb]), "v"(acc[1][jb]));
            continue;
        }
        if (kAdapt && warm_it) {
            // This wave's first row block, T still at its preset: the sums are looked at twice. First only to lower T (LDS) -- then the
            // workgroup exchanges with `smin` -- then, still in their registers, for the block's checks like any other block's. (Until
            // round 4 the block was WALKED twice: one row block in ~120 of a large launch, one in 15 of a one-pair launch over 1M rows.)
            if (active && full_block) {
#pragma unroll
                for (int jb = 0; jb < NJB; ++jb) {
                    const int q = jb * 16 + (lane & 15);
                    const float m2 = 2.0f * qinv_s[q];
                    float pv[8];
                    float mn = __builtin_huge_valf();
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const float gnv[4] = {gns[s].x, gns[s].y, gns[s].z, gns[s].w};
#pragma unroll
                        for (int reg = 0; reg < 4; ++reg) {
                            pv[4 * s + reg] = __builtin_fmaf(-m2, acc[s][jb][reg], gnv[reg]);
                            mn = fminf(mn, pv[4 * s + reg]);                   // NaN never enters, like k_gemm_tau's ordering
                        }
                    }
                    if (kSlots) {
                        // every lane's eight proxies lower their slots (LDS; the exchange turns the slots into T)
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const float tn = fmaxf((pv[i] + qn_s[q]) + win_s[q], 0.f);
                            if (pv[i] == pv[i] && tn < __uint_as_float(slot_s[q * 8 + i])) atomicMin(&slot_s[q * 8 + i], __float_as_uint(tn));
                        }
                    } else {
                        // the smallest proxy of the block's 32 rows lowers T (LDS)
                        float o = __shfl_xor(mn, 16, 64);
                        mn = o < mn ? o : mn;
                        o = __shfl_xor(mn, 32, 64);
                        mn = o < mn ? o : mn;
                        const float tn = fmaxf((mn + qn_s[q]) + win_s[q], 0.f);     // (NaN operands: fmaxf gives 0 only if both are NaN; a NaN tn fails the test below)
                        if (lane < 16 && tn < tau_s[q]) atomicMin((unsigned int*)&tau_s[q], __float_as_uint(tn));
                    }
                }
            }
            exchange_T();
        }
        if (!active) continue;
        if (full_block) {
            const float gmin = fminf(fminf(fminf(gns[0].x, gns[0].y), fminf(gns[0].z, gns[0].w)), fminf(fminf(gns[1].x, gns[1].y), fminf(gns[1].z, gns[1].w)));
            if (kAppend) {
                if (kDefer) {
                    pend = true;                                             // checked in the next row block's first step (or behind the loop)
                    if (kAdapt) {
#pragma unroll
                        for (int jb = 0; jb < NJB; ++jb) tqr[jb] = tq_s[jb * 16 + (lane & 15)];
                    }
                    p_rb = rb;
                    pg[0] = gns[0];
                    pg[1] = gns[1];
                    p_gmin = gmin;
                } else {
#pragma unroll
                    for (int jb = 0; jb < NJB; ++jb) check_jb(jb, rb, gns[0], gns[1], gmin);
                }
                continue;
            }
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb) {
                const int q = jb * 16 + (lane & 15);
                const float m2 = 2.0f * qinv_s[q];
                float pv[8];
                float mn = __builtin_huge_valf();
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const float gnv[4] = {gns[s].x, gns[s].y, gns[s].z, gns[s].w};
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        pv[4 * s + reg] = __builtin_fmaf(-m2, acc[s][jb][reg], gnv[reg]);
                        mn = fminf(mn, pv[4 * s + reg]);                   // NaN never enters, like k_gemm_tau's ordering
                    }
                }
                if (MODE == 2 && !sub_stride) {
                    smallest[jb] = fminf(smallest[jb], mn);
                } else if (MODE == 2) {
                    // the K-nearest sample: kRtSubsets = 8 positions (of a lane's eight rows) x 8 waves disjoint subsets -- every sampled row
                    // block feeds eight of them (a class of a few dozen rows then shows in eight subsets, not in one)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float v = pv[i] == pv[i] ? pv[i] : __builtin_huge_valf();
                        float o = __shfl_xor(v, 16, 64);
                        v = o < v ? o : v;
                        o = __shfl_xor(v, 32, 64);
                        v = o < v ? o : v;
                        if (lane < 16 && v < __builtin_huge_valf()) atomicMin(&smin[(size_t)(i * 8 + (wave & 7)) * sub_stride + q], fir::f32_orderable(v));
                    }
                } else {
                    float o = __shfl_xor(mn, 16, 64);
                    mn = o < mn ? o : mn;
                    o = __shfl_xor(mn, 32, 64);
                    mn = o < mn ? o : mn;
                    if (MODE == 0) { if (lane < 16) sample[(size_t)(rb - rb_begin) * (2 * kQT) + q] = mn; }
                }
            }
            continue;
        }
        // a block that straddles the end of the rows (or of the sample): row by row
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            const int q = jb * 16 + (lane & 15);
            const float m2 = 2.0f * qinv_s[q];
            const float tq = kAdapt ? tq_s[q] : tau_s[q];
            float mn = __builtin_huge_valf();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t row = rb * 32 + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3);
                if (row >= n || (MODE != 2 && (row < row_begin || row >= row_end))) continue;
                const float p = __builtin_fmaf(-m2, acc[i >> 2][jb][i & 3], gnorm[row]);
                if (MODE == 0) {
                    if (row < sample_rows) mn = p < mn ? p : mn;
                } else if (MODE == 2) {
                    mn = p < mn ? p : mn;
                } else if (p < tq) {
                    const int slot = atomicAdd(&counts[q], 1);
                    if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(p, (uint32_t)row);
                }
            }
            if (MODE == 2 && !sub_stride) {
                smallest[jb] = fminf(smallest[jb], mn);
            } else if (MODE != 1 && !kAdapt) {
                float o = __shfl_xor(mn, 16, 64);
                mn = o < mn ? o : mn;
                o = __shfl_xor(mn, 32, 64);
                mn = o < mn ? o : mn;
                if (MODE == 0) { if (lane < 16) sample[(size_t)(rb - rb_begin) * (2 * kQT) + q] = mn; }
                else if (lane < 16 && mn < __builtin_huge_valf()) atomicMin(&smin_blk[q], fir::f32_orderable(mn));     // (a straddling block: one subset takes its minimum)
            }
        }
    }
    if (kDefer && pend) {
        // the last full row block of this wave (no first step followed it); check_jb is the row loop's lambda -- the same code, here
        // against the values the loop left behind
        auto last_jb = [&](int jb) {
            const int q = jb * 16 + (lane & 15);
            const float m2 = 2.0f * qinv_s[q];
            const float tq = kAdapt ? tq_s[q] : tau_s[q];
            const float gnv[8] = {pg[0].x, pg[0].y, pg[0].z, pg[0].w, pg[1].x, pg[1].y, pg[1].z, pg[1].w};
            float pv[8];
            float mn = __builtin_huge_valf();
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                pv[i] = __builtin_fmaf(-m2, acc[i >> 2][jb][i & 3], gnv[i]);
                mn = fminf(mn, pv[i]);
            }
            if (mn < tq) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (pv[i] < tq) {
                        const int64_t row = p_rb * 32 + 16 * (i >> 2) + 4 * (lane >> 4) + (i & 3);
                        const unsigned long long key = fir::key_pack(pv[i], (uint32_t)row);
                        const int st = atomicAdd(&scnt[q], 1);
                        if (st < kXStage) {
                            skeys[q * kXStage + st] = key;
                        } else {
                            const int slot = atomicAdd(&counts[q], 1);
                            if (slot < kListCap) lists[(size_t)q * kListCap + slot] = key;
                        }
                    }
                }
                if (kAdapt) lower_T(q, pv, mn);
            }
        };
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) last_jb(jb);
    }
    if (kAppend) {
        __syncthreads();                             // every wave's staged appends are in
        if (threadIdx.x < 2 * kQT) {
            const int q = threadIdx.x;
            const int cnt = scnt[q] < kXStage ? scnt[q] : kXStage;
            if (cnt > 0) {
                const int base = atomicAdd(&counts[q], cnt);
                for (int st = 0; st < cnt; ++st)
                    if (base + st < kListCap) lists[(size_t)q * kListCap + base + st] = skeys[q * kXStage + st];
            }
        }
    }
    if (MODE == 2 && !sub_stride) {
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
            float v = smallest[jb];
            v = fminf(v, __shfl_xor(v, 16, 64));
            v = fminf(v, __shfl_xor(v, 32, 64));
            if (lane < 16 && v < __builtin_huge_valf()) atomicMin(&smin[jb * 16 + lane], fir::f32_orderable(v));
        }
    }
#undef FIR_X_BLOCK
#undef FIR_X_LD
}

// k_gemm_scan_f16 (the one-to-eight-query nomination scan, below in fir_gemm.hip) on the 16-row fragment order: lane l of
// piece 2 kk + s holds row 16 s + (l & 15), feature quarter l >> 4 of step kk.
typedef _Float16 f16x2x __attribute__((ext_vector_type(2)));
template <int NQ>
__global__ void __launch_bounds__(256) k_gemm_scan_f16x(const uint4* __restrict__ gh, const float* __restrict__ gnorm, const uint4* __restrict__ qh,
                                                         const float* __restrict__ qinv, int64_t n, int dk16, float* __restrict__ proxies,
                                                         unsigned int* __restrict__ smin) {
    extern __shared__ __attribute__((aligned(16))) uint4 qsx[];           // [step][feature quarter][query]
    const int dk32 = dk16 >> 1;
    for (int idx = threadIdx.x; idx < dk32 * 4 * NQ; idx += blockDim.x) {
        const int i = idx % NQ, qt = (idx / NQ) & 3, kk = idx / (4 * NQ);
        qsx[idx] = qh[(size_t)kk * 64 + 16 * qt + i];                     // queries 0..7 sit in query block 0 of the pair's fragments
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, qt = lane >> 4;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nrb = (n + 31) / 32;
    float m2[NQ], smallest[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { m2[i] = 2.0f * qinv[i]; smallest[i] = __builtin_huge_valf(); }
    for (int64_t rb = gw; rb < nrb; rb += nw) {
        const uint4* a = gh + (size_t)rb * dk16 * 64 + lane;
        float acc[2][NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc[0][i] = acc[1][i] = 0.f;
        for (int kb0 = 0; kb0 < dk16; kb0 += 8) {                          // dk16 is a multiple of 8
            uint4 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = ld_nt(a + (size_t)(kb0 + u) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f16x2x gv[4];
                __builtin_memcpy(gv, &g[u], 16);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const uint4 qq = qsx[(((kb0 + u) >> 1) * 4 + qt) * NQ + i];
                    f16x2x qv[4];
                    __builtin_memcpy(qv, &qq, 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[u & 1][i] = __builtin_amdgcn_fdot2(gv[t], qv[t], acc[u & 1][i], false);
                }
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int64_t row = rb * 32 + 16 * s + (lane & 15);
            const float gn = row < n ? gnorm[row] : 0.f;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                float dot = acc[s][i] + __shfl_xor(acc[s][i], 16, 64);    // the four feature quarters of the row
                dot += __shfl_xor(dot, 32, 64);
                if (lane < 16 && row < n) {
                    const float p = __builtin_fmaf(-m2[i], dot, gn);
                    proxies[(size_t)i * n + row] = p;
                    smallest[i] = fminf(smallest[i], p);                   // NaN never enters
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float v = smallest[i];
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
        if (lane == 0 && v < __builtin_huge_valf()) atomicMin(&smin[i], fir::f32_orderable(v));
    }
}
