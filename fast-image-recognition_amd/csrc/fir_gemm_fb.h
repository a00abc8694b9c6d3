// fir_gemm_fb.h -- what happens to a query whose certificate did not hold, entirely on the device and in stream order
// (included by fir_gemm.hip, inside its anonymous namespace).
//
// Rounds 1-3 read the certificate flags back (one stream synchronisation per call) and sent every uncertified query through
// the exact streaming scan from the host. That made the "asynchronous" device-pointer calls synchronous whenever the matrix
// cores answered, and it made a loose append threshold expensive: a list that overflows (kListCap rows below the threshold)
// cost a host round trip plus ~0.35 ms of exact scan per 8 queries. Now:
//
//   1. the re-rank kernels append every uncertified query to `list` (its length: state[0]) together with the bound a SECOND pass
//      may use for it: tau2 = min(first bound, smallest stored proxy + one window) -- at or above (smallest proxy of ALL rows
//      + window), so the second list holds every possible winner, and as tight as the first pass can know (for the adaptive
//      flow the first bound already is exactly that);
//   2. up to kScRounds "second chance" rounds take 128 of those queries each through ONE more matrix-core pass (the sample-flow
//      kernel k_gemm_proxy_f16x<1, *> with tau2, every CU on the one pair) and the same re-rank + certificate; every kernel of
//      a round returns at its first instruction when the list has nothing for it, so the usual call pays a few empty launches;
//   3. what is still uncertified (NaN / infinite operands, more exact ties than a list holds) is collected again
//      (state[1]) and answered by k_gemm_exact_fb: the reference's own arithmetic over all rows, 8 queries per read of the
//      tiled f32 gallery, looping on the device over however many queries there are; K rounds for the K nearest rows.
//
// No host synchronisation anywhere: the keys are final in stream order. The counters are running totals the statistics
// entry points read (fir_gemm_stats).
#pragma once

constexpr int kScQueries = 2 * kQT;      // queries of one second-chance round (one pair of passes)
constexpr int kScRounds = 4;             // second-chance rounds per call at most; the rest of a longer list goes to the exact scan

// device words of one fir_gemm (int[kFbStateWords]): [0] count, [1] count2 -- cleared at the start of every call --, [2] notes taken,
// [4..5] and [6..7]: running totals (64-bit) of queries that took a second pass / the exact device scan, [8..39]: four floats for each
// of the first eight uncertified queries of the state's life (fir_gemm_uncertified_notes)
constexpr int kFbStateWords = 40;
__device__ __forceinline__ void fb_note(int* state, int cnt, float bound, float smallest, float qn) {
    if (state[2] < 8) {                                            // (racy on purpose: a cheap filter in front of the atomic)
        const int i = atomicAdd(&state[2], 1);
        if (i < 8) {
            float* o = (float*)(state + 8) + 4 * i;
            o[0] = (float)cnt; o[1] = bound; o[2] = smallest; o[3] = qn;
        }
    }
}
struct RerankFb {
    int* state;          // the state words above
    int* list;           // first pass: uncertified queries are appended here (index within the call)
    float* tau2;         // first pass: [query of the call] the bound the second-chance pass appends below
    int q_base;          // first pass: index within the call of block 0's query
    const int* qmap;     // second chance: block b re-ranks scratch slot b for query qmap[b] of the call ...
    int live_off;        // ... while b < state[0] - live_off
};

// Second chance, per slot (one wave each): what k_gemm_qprep_f16 computes, for query list[live_off + slot] of the call; the
// bound is tau2 of that query, the padding slots of a half-filled pair append nothing (-inf).
__global__ void __launch_bounds__(64) k_gemm_sc_prep(const float* __restrict__ queries, int d, int qstride, int gallery_exp, const int* __restrict__ state,
                                                      const int* __restrict__ list, int live_off, const float* __restrict__ tau2_all,
                                                      float* __restrict__ qnorm, float* __restrict__ qmul, float* __restrict__ qinv, float* __restrict__ tau,
                                                      int* __restrict__ counts) {
    const int slot = blockIdx.x;
    int live = state[0] - live_off;
    if (live <= 0) return;                                          // (uniform: nothing of this round runs)
    live = live < kScQueries ? live : kScQueries;
    const bool valid = slot < live;
    const int q = valid ? list[live_off + slot] : 0;
    float s = 0.f, m = 0.f;
    bool bad = false;
    if (valid)
        for (int k = threadIdx.x; k < d; k += 64) {
            const float x = queries[(size_t)q * qstride + k];
            s += x * x;
            m = fmaxf(m, fabsf(x));
            bad = bad || !(fabsf(x) < __builtin_huge_valf());
        }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        s += __shfl_xor(s, off, 64);
        m = fmaxf(m, __shfl_xor(m, off, 64));
    }
    bad = __any(bad);
    if (threadIdx.x == 0) {
        int ex = 0;
        if (m > 0.f) (void)frexpf(m, &ex);
        const int sh = m > 0.f ? 14 - ex : 0;
        if (sh < -100 || sh > 100 || gallery_exp < -100 || gallery_exp > 100) bad = true;
        qnorm[slot] = (bad && valid) ? __builtin_nanf("") : s;
        qmul[slot] = bad ? 0.f : ldexpf(1.0f, sh);
        qinv[slot] = !valid ? 0.f : bad ? __builtin_nanf("") : ldexpf(1.0f, -sh - gallery_exp);
        tau[slot] = valid ? tau2_all[q] : -__builtin_huge_valf();
        counts[slot] = 0;
    }
}

// After the second-chance rounds: the queries of `list` that are still uncertified -> list2 (state[1]), their K keys preset for
// the exact scan's atomic minima; the running totals move on. One workgroup.
__global__ void __launch_bounds__(256) k_gemm_fb_collect(int* __restrict__ state, const int* __restrict__ list, const int* __restrict__ ok,
                                                          int* __restrict__ list2, unsigned long long* __restrict__ keys, int k) {
    __shared__ int cnt_s;
    const int count = state[0];
    if (count == 0) return;                                          // (state[1] was cleared with state[0])
    if (threadIdx.x == 0) cnt_s = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < count; i += 256) {
        const int q = list[i];
        if (!ok[q]) {
            const int slot = atomicAdd(&cnt_s, 1);
            list2[slot] = q;
            for (int r = 0; r < k; ++r) keys[(size_t)q * k + r] = kKeyNone;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        state[1] = cnt_s;
        unsigned long long* tot = (unsigned long long*)(state + 4);
        tot[0] += (unsigned long long)count;
        tot[1] += (unsigned long long)cnt_s;
    }
}

// The exact scan of the uncertified rest: db_features.cpp:22-42 (sequential, un-fused f32 sum over the compared features, one
// division) for every row, 8 queries per read of the tiled gallery, first minimum on (distance, row) as packed keys
// (db_features.cpp:325-333); `round` r > 0 takes the smallest key above the query's key of round r - 1 (the K nearest rows in K
// launches: this path is rare, it only has to be right). Loops on the device over all state[1] queries.
// Dynamic LDS: dp4 * 4 * 8 floats, [feature][query].
__global__ void __launch_bounds__(256) k_gemm_exact_fb(const float4* __restrict__ gal4, int64_t n, int dp4, int d, int64_t row_offset,
                                                        const float* __restrict__ queries, int qstride, const int* __restrict__ state,
                                                        const int* __restrict__ list2, unsigned long long* __restrict__ keys, int k, int round) {
    extern __shared__ __attribute__((aligned(16))) float qs_fb[];
    const int cnt = state[1];
    if (cnt == 0) return;
    const int d4 = (d + 3) >> 2;
    const int lane = threadIdx.x & 63;
    const int64_t tiles = (n + 63) / 64;
    const int gw = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int t8 = 0; t8 * 8 < cnt; ++t8) {
        const int nq = cnt - t8 * 8 < 8 ? cnt - t8 * 8 : 8;
        __syncthreads();                                             // everyone has left the previous tile of queries
        for (int i = threadIdx.x; i < d4 * 4 * 8; i += 256) {
            const int qi = i & 7, kf = i >> 3;
            qs_fb[i] = (qi < nq && kf < d) ? queries[(size_t)list2[t8 * 8 + qi] * qstride + kf] : 0.f;
        }
        __syncthreads();
        unsigned long long prev[8], best[8];
#pragma unroll
        for (int qi = 0; qi < 8; ++qi) {
            prev[qi] = (round > 0 && qi < nq) ? keys[(size_t)list2[t8 * 8 + qi] * k + round - 1] : 0ull;
            best[qi] = kKeyNone;
        }
        for (int64_t t = gw; t < tiles; t += nw) {
            const float4* p = gal4 + (size_t)t * dp4 * 64 + lane;
            float acc[8];
#pragma unroll
            for (int qi = 0; qi < 8; ++qi) acc[qi] = 0.f;
            for (int c = 0; c < d4; ++c) {
                const float4 g = p[(size_t)c * 64];
                const float gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 qa = *(const float4*)&qs_fb[(c * 4 + j) * 8], qb = *(const float4*)&qs_fb[(c * 4 + j) * 8 + 4];
                    acc[0] = fir::accum<fir::kL2>(acc[0], qa.x, gv[j]);
                    acc[1] = fir::accum<fir::kL2>(acc[1], qa.y, gv[j]);
                    acc[2] = fir::accum<fir::kL2>(acc[2], qa.z, gv[j]);
                    acc[3] = fir::accum<fir::kL2>(acc[3], qa.w, gv[j]);
                    acc[4] = fir::accum<fir::kL2>(acc[4], qb.x, gv[j]);
                    acc[5] = fir::accum<fir::kL2>(acc[5], qb.y, gv[j]);
                    acc[6] = fir::accum<fir::kL2>(acc[6], qb.z, gv[j]);
                    acc[7] = fir::accum<fir::kL2>(acc[7], qb.w, gv[j]);
                }
            }
            const int64_t row = t * 64 + lane;
            if (row < n) {
#pragma unroll
                for (int qi = 0; qi < 8; ++qi) {
                    const float dist = acc[qi] / (float)d;                       // db_features.cpp:40
                    if (dist < fir::kNotFound) {                                // (false for NaN: such a row never wins)
                        const unsigned long long key = fir::key_pack(dist, (uint32_t)(row + row_offset));
                        if ((round == 0 || key > prev[qi]) && key < best[qi]) best[qi] = key;
                    }
                }
            }
        }
#pragma unroll
        for (int qi = 0; qi < 8; ++qi) {
            const unsigned long long mk = fir::wave_min_u64(best[qi]);
            if (lane == 0 && qi < nq && mk != kKeyNone) atomicMin(&keys[(size_t)list2[t8 * 8 + qi] * k + round], mk);
        }
    }
}
