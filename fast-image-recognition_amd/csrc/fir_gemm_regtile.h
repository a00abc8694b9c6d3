// fir_gemm_regtile.h -- the full pass of the fp16 matrix-core path with the QUERY fragments in registers and the gallery
// fragments streamed through an LDS ring by LDS-DMA (included by fir_gemm.hip, inside its anonymous namespace).
//
// k_gemm_proxy_f16<1, *> keeps the 128-query tile in LDS and streams the gallery through VGPRs: with two 246-register
// waves per SIMD a wave can keep only one 8 KiB unit of gallery loads in flight, and rocprofv3 showed the waves parked in
// s_waitcnt for 31 % of their cycles with the matrix pipe 53 % busy (profiles/r02_rocprofv3_pmc_gemm_f16_share8_v1.json).
// Here the roles are swapped:
//   * a workgroup is 8 waves, a COMPUTE wave and a LOADER wave on every SIMD (<= 256 registers each); compute wave w owns
//     queries 32w .. 32w+31 of the 128-query tile and holds their fragments for ALL k-blocks in registers (4 VGPRs per
//     k-block: 128 at d = 512);
//   * the gallery fragments of a row block (32 rows) are fetched ONCE per workgroup by the loader waves with
//     global_load_lds_dwordx4 -- no VGPRs, asynchronous -- into a ring of eight 16 KiB chunks (128 KiB of the CU's 160
//     KiB), seven chunks ahead of the one being consumed, and every compute wave reads every chunk from LDS (ds_read_b128,
//     three k-blocks ahead of its MFMAs); issuing an LDS-DMA request costs the issuing wave 60-180 cycles
//     (MI355X_MICROARCH.md), which is why the compute waves do not do it themselves (measured: 535k instead of 870k
//     queries/s when they did); one s_barrier per chunk publishes the loaders' pieces and frees the slot the next
//     request overwrites;
//   * one accumulator tile per wave (32 rows x 32 queries); the epilogue of row block r (p = |g|^2 - 2 q.g, running
//     minimum, the rare append) is interleaved with the first MFMAs of row block r + 1; the row norms ride the same DMA
//     queue into a small LDS ring;
//   * appends go to a per-workgroup staging area in LDS (LDS atomics count in lgkmcnt): a returning GLOBAL atomic would
//     have to wait for every older DMA in the in-order vmcnt queue, i.e. drain the prefetch. The staging area is flushed
//     to the global lists once, at the end (a query that overflows its 16 staged entries appends directly: rare, correct).
// The LDS-DMA is issued from inline assembly: the compiler orders every ds_read behind a pending LDS-DMA it knows about
// with s_waitcnt vmcnt(0) (it cannot see that ring slots do not alias), which would serialise the stream; the waits that
// are needed are written out (s_waitcnt vmcnt(N) counts this wave's younger requests).
#pragma once

constexpr int kRtRing = 8;             // LDS ring slots (chunks)
constexpr int kRtStage = 16;           // staged appends per query and workgroup
constexpr int kRtNormSlots = 16;       // row blocks whose norms are resident
#ifndef FIR_RT_DEPTH
#define FIR_RT_DEPTH 4
#endif
constexpr int kRtDepth = FIR_RT_DEPTH;  // LDS reads in flight per compute wave (k-blocks ahead of the MFMA)

template <int DKB>
struct RegTile {
    static constexpr int CK = (DKB % 16 == 0) ? 16 : 8;      // k-blocks per chunk
    static constexpr int CPR = DKB / CK;                      // chunks per row block
    static constexpr int P = CK / 4;                          // DMA pieces (1 KiB each) per loader wave and chunk
    static constexpr int kAhead = (kRtRing - 3) * P;          // a loader's requests younger than the chunk it publishes at a barrier
    static constexpr size_t ring_bytes = (size_t)kRtRing * CK * 1024;
    static constexpr size_t norm_bytes = (size_t)kRtNormSlots * 64 * sizeof(float);
    static constexpr size_t stage_bytes = (size_t)128 * kRtStage * 8 + 128 * sizeof(int);
    static constexpr size_t lds_bytes = ring_bytes + norm_bytes + stage_bytes;
};

template <bool NT>
__device__ __forceinline__ void rt_dma16(const uint4* g, uint32_t lds_byte) {               // 64 lanes x 16 B -> LDS [lds_byte, +1 KiB)
    if (NT) asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off nt" ::"s"(lds_byte), "v"(g) : "memory");
    else asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(lds_byte), "v"(g) : "memory");
}
__device__ __forceinline__ void rt_dma4(const float* g, uint32_t lds_byte) {                // 64 lanes x 4 B -> LDS [lds_byte, +256 B)
    asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dword %1, off" ::"s"(lds_byte), "v"(g) : "memory");
}

#ifndef FIR_RT_OPT
#define FIR_RT_OPT 1           // bit 0: p by one fma; bit 1: epilogue arithmetic after the MFMAs of the first chunk instead of between them
#endif
#ifndef FIR_RT_DBG
#define FIR_RT_DBG 0           // experiment builds only (results are then meaningless): 1 no appends, 2 no MFMAs, 4 no DMA requests
#endif

// SAMPLE = false: the full pass -- rows [0, rows) of the gallery, every row with p < tau[q] is appended to the query's list.
// SAMPLE = true: the sample pass -- rows / 32 row blocks, every rb_stride-th one of the gallery (spread over all of it: the
// reference's galleries are ordered by class, a prefix would sample the first classes only), smin[q] <- the smallest proxy seen
// (as fir::f32_orderable bits, by atomicMin; the caller presets +inf): all the threshold needs (k_gemm_tau_min).
template <int DKB, bool SAMPLE>
__global__ void __launch_bounds__(512, 1) k_gemm_proxy_f16_regtile(const uint4* __restrict__ gh, const float* __restrict__ gnorm, const uint4* qh,
                                                                    const float* __restrict__ qinv, int64_t n, int64_t rows, int rb_stride, const float* tau,
                                                                    unsigned long long* lists, int* counts, unsigned int* smin, int share, int nt,
                                                                    int sub_stride = 0) {
    // sub_stride (SAMPLE, top-K): the sample is kept as kRtSubsets disjoint subsets' minima, smin[subset * sub_stride + query] -- the K-th
    // smallest of them is at or above the K-th smallest proxy of all rows (k_gemm_tau_kmin). 0: one minimum per query (top-1).
    using RT = RegTile<DKB>;
    constexpr int CK = RT::CK, CPR = RT::CPR, P = RT::P;
    constexpr int dbg = FIR_RT_DBG;
    extern __shared__ __attribute__((aligned(16))) uint4 rt_lds[];
    uint4* ring = rt_lds;
    float* nring = (float*)((char*)rt_lds + RT::ring_bytes);
    unsigned long long* skeys = (unsigned long long*)((char*)rt_lds + RT::ring_bytes + RT::norm_bytes);
    int* scnt = (int*)(skeys + 128 * kRtStage);
    const uint32_t ring_base = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)ring;
    const uint32_t nring_base = (uint32_t)(uintptr_t)(void __attribute__((address_space(3)))*)nring;
    // which pair of passes and which row blocks: as k_gemm_proxy_f16 (share = pairs that walk the same rows together)
    const int w = (int)blockIdx.x, xcd = w & 7, slot_wg = w >> 3;
    const int ranges = ((int)gridDim.x >> 3) / share * 8;
    const int range = xcd + 8 * (slot_wg / share);
    if (range >= ranges) return;                                   // uniform per workgroup
    const size_t pr = (size_t)(slot_wg % share);
    qh += pr * 4 * DKB * 64;
    qinv += pr * 2 * kQT;
    tau += pr * 2 * kQT;
    lists += pr * 2 * kQT * kListCap;
    counts += pr * 2 * kQT;
    smin += pr * 2 * kQT;
    const int64_t nrb_all = (rows + 31) / 32;
    const int64_t rb_first = nrb_all * range / ranges, rb_last = nrb_all * (range + 1) / ranges;
    const int nrb = (int)(rb_last - rb_first);
    if (nrb <= 0) return;
    const int lane = threadIdx.x & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);     // 0-3: compute waves, 4-7: loader waves (one of each per SIMD)
    if (threadIdx.x < 128) scnt[threadIdx.x] = 0;
    const int total = nrb * CPR;                                   // chunks of this workgroup
    if (wave8 >= 4) {
        // ---- loader: keeps kRtRing - 1 chunks requested ahead of the one the compute waves are on ----
        const int lw = wave8 - 4;
        auto issue = [&](int c) {                                  // this wave's pieces of chunk c (past the end: the last chunk again, into a slot nobody reads)
            if (dbg & 4) return;
            const int cc = c < total ? c : total - 1;
            const int i = cc / CPR, cp = cc - i * CPR;
            const uint4* src = gh + ((size_t)(rb_first + i) * rb_stride * DKB + (size_t)cp * CK) * 64 + lane;
            const uint32_t dst = ring_base + (uint32_t)(c & (kRtRing - 1)) * (CK * 1024);
            if (nt) {                                              // streamed once (one pair per launch): non-temporal
#pragma unroll
                for (int j = 0; j < P; ++j) rt_dma16<true>(src + (size_t)(lw + 4 * j) * 64, dst + (uint32_t)(lw + 4 * j) * 1024);
            } else {                                               // the pairs of the launch share the stream through L2
#pragma unroll
                for (int j = 0; j < P; ++j) rt_dma16<false>(src + (size_t)(lw + 4 * j) * 64, dst + (uint32_t)(lw + 4 * j) * 1024);
            }
            if (cp == 0 && c < total && lw == (i & 3)) {           // the row block's 32 squared norms (one loader asks, twice over the 64 lanes)
                int64_t row = (rb_first + i) * rb_stride * 32 + (lane & 31);
                row = row < n ? row : n - 1;
                rt_dma4(gnorm + row, nring_base + (uint32_t)(i & (kRtNormSlots - 1)) * 256);
            }
        };
        __syncthreads();                                           // (the compute waves' start-up barrier)
        for (int c = 0; c < kRtRing - 2; ++c) issue(c);
        for (int c = 0; c < total; ++c) {
            // barrier c: my pieces of chunk c have landed (the requests of the kRtRing - 3 younger chunks may still be in flight) ...
            asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(RT::kAhead) : "memory");
            // ... so have the other loaders'. The compute waves pass barrier c BEFORE they start on chunk c - 1 (they read one
            // chunk behind what has landed, so that their LDS reads run on across chunk borders): what they are done with is
            // chunk c - 2, and its slot takes chunk c + kRtRing - 2
            issue(c + kRtRing - 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // nothing of mine may land in LDS after the workgroup has gone
        return;
    }
    // ---- compute wave `wave8`: queries 32 * wave8 .. + 31 of the tile, fragments of all k-blocks in registers ----
    const int wave = wave8;
    uint4 B[DKB];
#pragma unroll
    for (int kb = 0; kb < DKB; ++kb) B[kb] = qh[((size_t)wave * DKB + kb) * 64 + lane];
    const int q = wave * 32 + (lane & 31);
    const float m2 = 2.0f * qinv[q];
    const float tq = SAMPLE ? 0.0f : (dbg & 1) ? -__builtin_huge_valf() : tau[q];
    float smallest = __builtin_huge_valf();                        // SAMPLE: running minimum of this lane's proxies
    float sm1 = __builtin_huge_valf(), sm2 = __builtin_huge_valf(), sm3 = __builtin_huge_valf();   // ... of row blocks 1, 2, 3 mod 4 (sub_stride != 0; `smallest` then: 0 mod 4)
    __syncthreads();                                               // staging counters zeroed
    f32x16 acc = {0.f}, prev = {0.f};
    // gallery fragments come out of LDS kRtDepth k-blocks ahead of the MFMA that uses them, across chunk borders: the wave
    // passes barrier c + 1 (chunk c + 1 has landed) before it starts on chunk c
    asm volatile("s_barrier" ::: "memory");                        // barrier 0
    uint4 af[kRtDepth];
    {
        const uint4* a = ring + lane;
#pragma unroll
        for (int j = 0; j < kRtDepth; ++j) af[j] = a[(size_t)j * 64];
    }
    for (int i = 0; i <= nrb; ++i) {                               // iteration nrb only finishes row block nrb - 1
        const bool have_cur = i < nrb, have_prev = i > 0;
        // epilogue of the previous row block, spread over this block's first chunk (3 vector ops per MFMA gap)
        float pv[16];
        float mn = __builtin_huge_valf();
        float4 g4[4];
        if (have_prev) {
            const float* gp = nring + ((i - 1) & (kRtNormSlots - 1)) * 64 + 4 * (lane >> 5);       // rows 8g + 4h + 0..3 of the block
#pragma unroll
            for (int g = 0; g < 4; ++g) g4[g] = *(const float4*)(gp + 8 * g);
        }
        const float gnv[16] = {g4[0].x, g4[0].y, g4[0].z, g4[0].w, g4[1].x, g4[1].y, g4[1].z, g4[1].w,
                               g4[2].x, g4[2].y, g4[2].z, g4[2].w, g4[3].x, g4[3].y, g4[3].z, g4[3].w};
#pragma unroll
        for (int cp = 0; cp < CPR; ++cp) {
            if (!have_cur) break;
            const int c = i * CPR + cp;
            if (c + 1 < total) asm volatile("s_barrier" ::: "memory");      // barrier c + 1: chunk c + 1 has landed; chunk c - 1 may be overwritten
            const uint4* a_cur = ring + (size_t)(c & (kRtRing - 1)) * CK * 64 + lane;
            const uint4* a_nxt = ring + (size_t)((c + 1 < total ? c + 1 : c) & (kRtRing - 1)) * CK * 64 + lane;
#pragma unroll
            for (int kb = 0; kb < CK; ++kb) {
                const uint4 a0 = af[kb % kRtDepth];
                af[kb % kRtDepth] = kb + kRtDepth < CK ? a_cur[(size_t)(kb + kRtDepth) * 64] : a_nxt[(size_t)(kb + kRtDepth - CK) * 64];
                __builtin_amdgcn_sched_barrier(0);                  // the read of k-block kb + kRtDepth stays ahead of the MFMA of kb
                if (!(dbg & 2)) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_f16x8(a0), as_f16x8(B[cp * CK + kb]), acc, 0, 0, 0);
                else acc[kb & 15] += __uint_as_float(a0.x);
                if (cp == 0 && have_prev && kb < 16 && !(FIR_RT_OPT & 2)) {
                    pv[kb] = (FIR_RT_OPT & 1) ? __builtin_fmaf(-m2, prev[kb], gnv[kb]) : gnv[kb] - m2 * prev[kb];
                    mn = fminf(mn, pv[kb]);                         // NaN never enters, like k_gemm_tau's ordering
                }
            }
            if (cp == 0 && have_prev && (CK < 16 || (FIR_RT_OPT & 2))) {   // 8-k-block chunks: the second half of the epilogue arithmetic
#pragma unroll
                for (int r = (FIR_RT_OPT & 2) ? 0 : CK; r < 16; ++r) {
                    pv[r] = (FIR_RT_OPT & 1) ? __builtin_fmaf(-m2, prev[r], gnv[r]) : gnv[r] - m2 * prev[r];
                    mn = fminf(mn, pv[r]);
                }
            }
        }
        if (have_prev) {
            if (!have_cur) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    pv[r] = (FIR_RT_OPT & 1) ? __builtin_fmaf(-m2, prev[r], gnv[r]) : gnv[r] - m2 * prev[r];
                    mn = fminf(mn, pv[r]);
                }
            }
            if (SAMPLE) {
                const int64_t rbs = (rb_first + i - 1) * rb_stride;
                if (rbs * 32 + 32 > n) {                            // the block that straddles the end of the gallery: its padding rows are no sample
                    mn = __builtin_huge_valf();
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (rbs * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2) < n) mn = fminf(mn, pv[r]);
                }
                if (sub_stride) {
                    const int sub = (i - 1) & 3;
                    smallest = sub == 0 ? fminf(smallest, mn) : smallest;
                    sm1 = sub == 1 ? fminf(sm1, mn) : sm1;
                    sm2 = sub == 2 ? fminf(sm2, mn) : sm2;
                    sm3 = sub == 3 ? fminf(sm3, mn) : sm3;
                } else {
                    smallest = fminf(smallest, mn);
                }
            } else if (__builtin_amdgcn_ballot_w64(mn < tq) != 0) { // rare (a few per cent of the row blocks): some lane holds a row below its query's tau
                const int64_t rb = rb_first + i - 1;
                unsigned hm = 0;                                    // this lane's rows below tau, bit r = accumulator register r
#pragma unroll
                for (int r = 0; r < 16; ++r) hm |= pv[r] < tq ? (1u << r) : 0u;
                if (rb * 32 + 32 > n) {                             // the block that straddles the end of the gallery: padding rows never qualify
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (rb * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2) >= n) hm &= ~(1u << r);
                }
                const int k = __popc(hm);
                int s0 = 0;
                if (k) s0 = atomicAdd(&scnt[q], k);                 // LDS: one reservation per lane, the slots of a lane are consecutive
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (__builtin_amdgcn_ballot_w64((hm >> r) & 1u) == 0) continue;      // wave-uniform: nobody has register r
                    if ((hm >> r) & 1u) {
                        const int64_t row = rb * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
                        const unsigned long long key = fir::key_pack(pv[r], (uint32_t)row);
                        const int s = s0 + __popc(hm & ((1u << r) - 1u));
                        if (s < kRtStage) {
                            skeys[q * kRtStage + s] = key;
                        } else {                                     // staging full (clusters of near-duplicates): straight to the list
                            const int gs = atomicAdd(&counts[q], 1);
                            if (gs < kListCap) lists[(size_t)q * kListCap + gs] = key;
                        }
                    }
                }
            }
        }
        prev = acc;
        acc = f32x16{0.f};
    }
    if (SAMPLE) {
        if (sub_stride) {
            // 16 row ranges x 4 row-block residues = kRtSubsets subsets (more ranges wrap around: unions of disjoint sets)
            float v[4] = {smallest, sm1, sm2, sm3};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float o = __shfl_xor(v[t], 32, 64);           // the two half-waves hold the same 32 queries
                v[t] = fminf(v[t], o);
                if (lane < 32 && v[t] < __builtin_huge_valf())
                    atomicMin(&smin[(size_t)(((range & 15) << 2) + t) * sub_stride + q], fir::f32_orderable(v[t]));
            }
            return;
        }
        const float o = __shfl_xor(smallest, 32, 64);               // the two half-waves hold the same 32 queries
        smallest = fminf(smallest, o);
        if (lane < 32 && smallest < __builtin_huge_valf()) atomicMin(&smin[q], fir::f32_orderable(smallest));
        return;
    }
    // flush the staged appends of this wave's 32 queries
    if (lane < 32) {
        const int cnt = scnt[q] < kRtStage ? scnt[q] : kRtStage;
        if (cnt > 0) {
            const int base = atomicAdd(&counts[q], cnt);
            for (int s = 0; s < cnt; ++s)
                if (base + s < kListCap) lists[(size_t)q * kListCap + base + s] = skeys[q * kRtStage + s];
        }
    }
}

// tau for the register-tile flow: the smallest SAMPLED proxy plus one rounding window (2.5 E d, as k_gemm_rerank's 2 E d
// with room). Every row that can still be the nearest has a proxy within 2 E d of the smallest proxy of ALL rows, which is
// not above the smallest sampled one: it is appended. And the certificate holds for the rest with room to spare: a row
// that was not appended has p >= tau, i.e. a reference distance >= (|q|^2 + p_s + 2.5 E d)/d - E, while the winner's is
// <= (|q|^2 + p_s)/d + E. About n / sample_rows + (rows inside the window) rows pass per query -- tens, not hundreds, which
// is what keeps the append path out of the full pass's way.
constexpr int kRtSubsets = 64;
// Top-K form of k_gemm_tau_min: the sample arrives as kRtSubsets disjoint subsets' minima per query; K of them are at or below the
// K-th smallest of these minima, so at least K rows of the gallery are: tau = that + one window appends every row that can be
// among the K nearest, and leaves room for the certificate of rank K. (With K << kRtSubsets the K smallest sampled proxies sit
// in different subsets almost always: the bound is within a rank or two of the K-th smallest of the whole sample.)
__global__ void __launch_bounds__(256) k_gemm_tau_kmin(const unsigned int* __restrict__ smin, int sub_stride, int k, float* __restrict__ tau,
                                                        int nq_total, int nq_valid, const float* __restrict__ qnorm,
                                                        const float* __restrict__ gnorm_max_p, float e_rel) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_total) return;
    if (q >= nq_valid) { tau[q] = -__builtin_huge_valf(); return; }
    unsigned int best[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) best[i] = 0xFFFFFFFFu;
    for (int s = 0; s < kRtSubsets; ++s) {
        unsigned int v = smin[(size_t)s * sub_stride + q];
        if (v == 0xFF800000u) continue;                                    // an empty subset (+inf preset)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const bool sw = v < best[i];
            const unsigned int t = best[i];
            best[i] = sw ? v : t;
            v = sw ? t : v;
        }
    }
    unsigned int kth = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < 8; ++i) kth = i == k - 1 ? best[i] : kth;
    float t = __builtin_huge_valf();                                      // fewer than K sampled subsets: everything passes, the list cap decides
    if (kth != 0xFFFFFFFFu) {
        const float v = fir::f32_from_orderable(kth) + 2.5f * e_rel * (qnorm[q] + gnorm_max_p[0]);
        t = v + fabsf(v) * 1e-6f + 1e-30f;
    }
    tau[q] = t;
}

__global__ void __launch_bounds__(256) k_gemm_tau_min(const unsigned int* __restrict__ smin, float* __restrict__ tau, int nq_total, int nq_valid,
                                                       const float* __restrict__ qnorm, const float* __restrict__ gnorm_max_p, float e_rel) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_total) return;
    if (q >= nq_valid) { tau[q] = -__builtin_huge_valf(); return; }       // padding queries of a half-filled pair: nothing is appended
    const unsigned int o = smin[q];
    float t = __builtin_huge_valf();                                      // no sampled proxy (NaN operands): everything passes, the list cap decides
    if (o != 0xFF800000u) {
        const float v1 = fir::f32_from_orderable(o);
        const float v = v1 + 2.5f * e_rel * (qnorm[q] + gnorm_max_p[0]);  // a NaN window gives a NaN tau: nothing passes, nothing is certified
        t = v + fabsf(v) * 1e-6f + 1e-30f;
    }
    tau[q] = t;
}
