// fir_gemm.hip -- large-batch L2 top-1 through the matrix cores, with the reference's exact answer.
//
// The streaming scan (fir_kernels.h) is HBM-bound at 8 queries per gallery pass and VALU-bound
// beyond. For big query batches the N x Qb x d work is a GEMM: p(q, g) = |g|^2 - 2 q.g orders the
// rows of one query exactly like the squared distance does, and q.g runs on MFMA. gfx950 has an
// f32-input MFMA (v_mfma_f32_32x32x2_f32: exact products, f32 fma chain) at the f32 vector peak
// (157 TF), i.e. ONE matrix op per (row, query, feature) instead of the scan's three VALU ops.
//
// The GEMM only NOMINATES rows; the answer is still the reference's:
//   1. proxies of a row sample give, per query, an upper bound tau of its C-th smallest proxy;
//   2. the full pass appends every row with p < tau to the query's candidate list;
//   3. every appended row whose proxy lies within the rounding window of the smallest one is re-ranked
//      with the reference's own arithmetic (sequential, un-fused f32: fir::accum<kL2>), first-minimum
//      tie-break on (distance, row);
//   4. a rigorous bound certifies the winner: every other row has p >= p_excl, hence a reference
//      distance >= (|q|^2 + p_excl)/d - E, with E covering every rounding on both sides; if that does
//      not exceed the winner's exact distance (the window reaches tau, the list overflowed, NaN), the
//      query is re-run through the exact streaming scan.
// So results are bit-identical to the scan path's -- index and distance -- by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"
#include "fir_common.h"

namespace {

using fir::kKeyNone;

#ifndef FIR_GEMM_BLOCK
#define FIR_GEMM_BLOCK 512
#endif
constexpr int kGemmBlock = FIR_GEMM_BLOCK;   // 512 = 8 waves: 2 per SIMD (one workgroup per CU: the query slab takes 128 KiB of LDS)
constexpr int kGemmMinBlocks = kGemmBlock <= 512 ? 2 : 1;
constexpr int kQT = 64;                // queries per pass (2 accumulator tiles of 32 per wave)
constexpr int kSlab8 = 64;             // query features staged in LDS at a time, in groups of 8 (512 features)
constexpr int kSlab16 = 32;            // bf16 variant: k-blocks of 16 features staged at a time (512 features)
constexpr int kPasses = 128;           // 64-query passes per super-batch at most (one set of launches, one set of scratch): 8192 queries
constexpr int kCand = 8;               // tau = the kCand-th smallest SAMPLED proxy (so ~kCand * n / sample rows get appended)
constexpr int kRerankGroup = 8;         // candidate rows staged in LDS at a time by the re-rank (fewer when rows are longer than ~4000 features)
constexpr size_t kRerankLdsMax = 144 * 1024;
constexpr int kListCap = 4096;         // appended (proxy, row) entries per query before "overflow"
constexpr int kMinSampleRows = 8192;   // rows whose proxies seed tau: max(8192, n / 64) -> ~512 appended rows per query

typedef float f32x16 __attribute__((ext_vector_type(16)));

thread_local char g_gemm_err[512];
int gemm_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_gemm_err, sizeof(g_gemm_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_gemm_err);
    return code;
}
#define GEMM_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return gemm_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,       \
                                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// MFMA operand layout shared by gallery and queries ("fragment order"): for a block of 32 vectors
// and a group kq of 8 features, lane l holds the float4
//   ( v[l&31][8kq+0+h], v[l&31][8kq+2+h], v[l&31][8kq+4+h], v[l&31][8kq+6+h] ),  h = l >> 5,
// so component i is the A (or B) operand of the MFMA that contracts features 8kq+2i, 8kq+2i+1
// (v_mfma_f32_32x32x2_f32: A[i = l&31][k = l>>5], B[k = l>>5][j = l&31]).
// tiled gallery (fir_kernels.h layout) -> fragment order + squared row norms.
__global__ void __launch_bounds__(256) k_gemm_pack_gallery(const float4* __restrict__ gal4, int64_t n, int dp4, int dq8,
                                                            float4* __restrict__ gm, float* __restrict__ gnorm) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;      // float4 index in gm
    const int64_t rblocks = (n + 31) / 32;
    if (o >= rblocks * dq8 * 64) return;
    const int l = (int)(o & 63);
    const int64_t t = o >> 6;
    const int kq = (int)(t % dq8);
    const int64_t rb = t / dq8;
    const int64_t row = rb * 32 + (l & 31);
    const int h = l >> 5;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (row < n) {
        const int64_t tile = row >> 6;
        const int r = (int)(row & 63);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 8 * kq + 2 * i + h;
            if (k < dp4 * 4) {
                const float4 g = gal4[(tile * dp4 + (k >> 2)) * 64 + r];
                v[i] = (k & 3) == 0 ? g.x : (k & 3) == 1 ? g.y : (k & 3) == 2 ? g.z : g.w;
            }
        }
    }
    gm[o] = make_float4(v[0], v[1], v[2], v[3]);
    if (kq == 0 && h == 0 && row < n) {      // one thread per row: its squared norm (any summation order: covered by E)
        const int64_t tile = row >> 6;
        const int r = (int)(row & 63);
        float s = 0.f;
        for (int c = 0; c < dp4; ++c) {
            const float4 g = gal4[(tile * dp4 + c) * 64 + r];
            s += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
        }
        gnorm[row] = s;
    }
}

// queries[nq][d] -> fragment order qm[jb][kq][lane] (float4), zero padded to kQT queries; qnorm[q] = |q|^2.
__global__ void __launch_bounds__(256) k_gemm_pack_queries(const float* q, int nq, int d, int dq8, float4* qm) {
    // blockIdx.y = pass: 64 queries each, packed back to back
    q += (size_t)blockIdx.y * kQT * d;
    nq -= (int)blockIdx.y * kQT;
    qm += (size_t)blockIdx.y * (kQT / 32) * dq8 * 64;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < (kQT / 32) * dq8 * 64) {
        const int l = o & 63;
        const int t = o >> 6;
        const int kq = t % dq8, jb = t / dq8;
        const int qi = jb * 32 + (l & 31), h = l >> 5;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (qi < nq) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int k = 8 * kq + 2 * i + h;
                if (k < d) v[i] = q[(size_t)qi * d + k];
            }
        }
        qm[o] = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// qnorm[q] = |q|^2 (one wave per query; any summation order is covered by the certificate's E).
__global__ void __launch_bounds__(64) k_gemm_qnorm(const float* __restrict__ q, int nq, int d, float* __restrict__ qnorm) {
    const int qi = blockIdx.x;
    float s = 0.f;
    if (qi < nq)
        for (int k = threadIdx.x; k < d; k += 64) s += q[(size_t)qi * d + k] * q[(size_t)qi * d + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off, 64);
    if (threadIdx.x == 0) qnorm[qi] = s;
}

// One wave: 32 gallery rows x 64 queries; p = |g|^2 - 2 q.g.
// MODE 0: store the proxies of rows < sample_rows into sample[q][row] (seeding tau).
// MODE 1: append (p, row) with p < tau[q] to the query's list.
// Dynamic LDS: the query tile in fragment order, 2 * dq8 * 64 float4.
template <int MODE>
__global__ void __launch_bounds__(kGemmBlock, kGemmMinBlocks) k_gemm_proxy(const float4* __restrict__ gm, const float* __restrict__ gnorm,
                                                               const float4* qm, int64_t n, int64_t row_begin,
                                                               int64_t row_end, int dq8, const float* tau, unsigned long long* lists, int* counts, float* sample,
                                                               int sample_rows) {
    extern __shared__ __attribute__((aligned(16))) float4 lq[];
    __shared__ float tau_s[kQT];
    {   // blockIdx.y = pass (64 queries each); every per-pass buffer is laid out pass-major
        const size_t ps = blockIdx.y;
        qm += ps * (kQT / 32) * dq8 * 64;
        tau += ps * kQT;
        lists += ps * kQT * kListCap;
        counts += ps * kQT;
        sample += ps * kQT * sample_rows;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;      // 8 waves for the full pass, 2 for the (short) sample pass so that it covers every CU
    // The query tile is staged in LDS one SLAB of features at a time (<= kSlab8 groups of 8 = 512 features, 128 KiB for
    // 64 queries); d <= 512 needs one slab, loaded once; longer vectors (d = 1280: 3 slabs) re-stage per row group.
    const int sq8 = dq8 < kSlab8 ? dq8 : kSlab8;
    const int nslab = (dq8 + sq8 - 1) / sq8;
    if (MODE == 1 && threadIdx.x < kQT) tau_s[threadIdx.x] = tau[threadIdx.x];
    const int64_t rb_begin = row_begin / 32, rb_end = (row_end + 31) / 32;
    const int64_t nrg = (rb_end - rb_begin + wpb - 1) / wpb;            // row groups of wpb x 32 rows; uniform trip count per block
    bool staged = false;
    for (int64_t rg = blockIdx.x; rg < nrg; rg += gridDim.x) {
        const int64_t rb = rb_begin + rg * wpb + wave;
        const bool active = rb < rb_end;
        f32x16 acc0 = {0.f}, acc1 = {0.f};
        for (int sl = 0; sl < nslab; ++sl) {
            const int k0 = sl * sq8;
            const int kw = dq8 - k0 < sq8 ? dq8 - k0 : sq8;            // groups in this slab (a multiple of 4)
            if (nslab > 1 || !staged) {
                __syncthreads();                                          // everyone is done with the previous slab
                for (int i = threadIdx.x; i < 2 * kw * 64; i += blockDim.x) {
                    const int jb = i / (kw * 64), r = i - jb * kw * 64;
                    lq[(size_t)jb * sq8 * 64 + r] = qm[((size_t)jb * dq8 + k0) * 64 + r];
                }
                __syncthreads();
                staged = true;
            }
            if (!active) continue;
            const float4* a = gm + ((size_t)rb * dq8 + k0) * 64 + lane;
            // kw is a multiple of 4: EIGHT gallery fragments in flight (named registers, no runtime indexing), each
            // re-issued right after its 8 MFMAs, i.e. 7 steps = 3 584 MFMA cycles (x2 with two waves per SIMD) ahead of
            // its use -- with 2 waves per SIMD the loads must cover the HBM latency on their own.
            const int last = kw - 1;
            float4 a0 = a[0], a1 = a[(size_t)(1 < last ? 1 : last) * 64], a2 = a[(size_t)(2 < last ? 2 : last) * 64],
                   a3 = a[(size_t)(3 < last ? 3 : last) * 64], a4 = a[(size_t)(4 < last ? 4 : last) * 64],
                   a5 = a[(size_t)(5 < last ? 5 : last) * 64], a6 = a[(size_t)(6 < last ? 6 : last) * 64],
                   a7 = a[(size_t)(7 < last ? 7 : last) * 64];
            // query fragments come from LDS one step AHEAD of the MFMAs that use them (b0/b1 = this step, n0/n1 = next)
            float4 b0 = lq[lane], b1 = lq[(size_t)sq8 * 64 + lane];
#define FIR_GEMM_STEP(AV, KQ)                                                              \
            {                                                                              \
                const int kn = (KQ) + 1 < kw ? (KQ) + 1 : (KQ);                            \
                const float4 n0 = lq[(size_t)kn * 64 + lane];                              \
                const float4 n1 = lq[(size_t)(sq8 + kn) * 64 + lane];                      \
                __builtin_amdgcn_sched_barrier(0); /* keep the two ds_reads up here */    \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, b0.x, acc0, 0, 0, 0);    \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.x, b1.x, acc1, 0, 0, 0);    \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, b0.y, acc0, 0, 0, 0);    \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.y, b1.y, acc1, 0, 0, 0);    \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, b0.z, acc0, 0, 0, 0);    \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.z, b1.z, acc1, 0, 0, 0);    \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, b0.w, acc0, 0, 0, 0);    \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(AV.w, b1.w, acc1, 0, 0, 0);    \
                b0 = n0;                                                                   \
                b1 = n1;                                                                   \
            }
            for (int kq = 0; kq < kw; kq += 8) {
                // (the tail re-loads clamp to the last fragment: harmless L2 hits, keeps the loop branch-free)
#define FIR_GEMM_NEXT(AV, OFF) AV = a[(size_t)(kq + (OFF) + 8 < kw ? kq + (OFF) + 8 : last) * 64];
                FIR_GEMM_STEP(a0, kq)
                FIR_GEMM_NEXT(a0, 0)
                FIR_GEMM_STEP(a1, kq + 1)
                FIR_GEMM_NEXT(a1, 1)
                FIR_GEMM_STEP(a2, kq + 2)
                FIR_GEMM_NEXT(a2, 2)
                FIR_GEMM_STEP(a3, kq + 3)
                FIR_GEMM_NEXT(a3, 3)
                if (kq + 4 < kw) {     // kw is a multiple of 4, not necessarily of 8
                    FIR_GEMM_STEP(a4, kq + 4)
                    FIR_GEMM_NEXT(a4, 4)
                    FIR_GEMM_STEP(a5, kq + 5)
                    FIR_GEMM_NEXT(a5, 5)
                    FIR_GEMM_STEP(a6, kq + 6)
                    FIR_GEMM_NEXT(a6, 6)
                    FIR_GEMM_STEP(a7, kq + 7)
                    FIR_GEMM_NEXT(a7, 7)
                }
#undef FIR_GEMM_NEXT
            }
#undef FIR_GEMM_STEP
        }
        if (!active) continue;
        // this wave's 32 squared row norms, one per lane (both halves), handed out by shuffles in the epilogue
        const int64_t nrow = rb * 32 + (lane & 31);
        const float gn_lane = nrow < n ? gnorm[nrow] : 0.0f;
        // C/D layout of the 32x32 MFMA: column (query) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int q = jb * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int roff = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int64_t row = rb * 32 + roff;
                const float gn = __shfl(gn_lane, roff, 64);     // every lane takes part (before any divergence)
                if (row >= n || row < row_begin || row >= row_end) continue;
                const float dot = jb == 0 ? acc0[reg] : acc1[reg];
                const float p = gn - 2.0f * dot;
                if (MODE == 0) {
                    if (row < sample_rows) sample[(size_t)q * sample_rows + row] = p;
                } else if (p < tau_s[q]) {
                    const int slot = atomicAdd(&counts[q], 1);
                    if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(p, (uint32_t)row);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// bf16-split variant. x = hi + lo + r with hi = bf16(x), lo = bf16(x - hi), |r| <= 2^-18 |x|: the dot product
// is taken as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 (bf16 products are exact in f32; 16x the f32-MFMA
// rate, so three of them are still 5.3x faster); the dropped lo.lo and r terms are <= 3 * 2^-18 |g||q| and go
// into the certificate's E. Fragment order for this MFMA: lane l (r = l & 31, h = l >> 5) holds the 8 features
// 16 kb + 8 h .. + 7 of vector r of its 32-block as 8 bf16 (one uint4); hi and lo fragments are stored side by side.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split_bf16(float x, unsigned short& hi, unsigned short& lo) {
    const __bf16 h = (__bf16)x;                    // round to nearest even
    const __bf16 l = (__bf16)(x - (float)h);
    __builtin_memcpy(&hi, &h, 2);
    __builtin_memcpy(&lo, &l, 2);
}

// tiled f32 gallery -> gb[(rb * dk16 + kb) * 2 + {hi, lo}][lane] (uint4 = 8 bf16)
__global__ void __launch_bounds__(256) k_gemm_pack_gallery_bf16(const float4* __restrict__ gal4, int64_t n, int dp4, int dk16,
                                                                 uint4* __restrict__ gb) {
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (rb, kb, lane)
    const int64_t rblocks = (n + 31) / 32;
    if (o >= rblocks * dk16 * 64) return;
    const int l = (int)(o & 63);
    const int64_t t = o >> 6;
    const int kb = (int)(t % dk16);
    const int64_t rb = t / dk16;
    const int64_t row = rb * 32 + (l & 31);
    const int h = l >> 5;
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * kb + 8 * h + j;
        float x = 0.f;
        if (row < n && k < dp4 * 4) {
            const float4 g = gal4[((row >> 6) * dp4 + (k >> 2)) * 64 + (row & 63)];
            x = (k & 3) == 0 ? g.x : (k & 3) == 1 ? g.y : (k & 3) == 2 ? g.z : g.w;
        }
        split_bf16(x, hi[j], lo[j]);
    }
    uint4 vh, vl;
    __builtin_memcpy(&vh, hi, 16);
    __builtin_memcpy(&vl, lo, 16);
    gb[(size_t)(t * 2) * 64 + l] = vh;
    gb[(size_t)(t * 2 + 1) * 64 + l] = vl;
}

// queries -> qb[(jb * dk16 + kb) * 2 + {hi, lo}][lane], zero padded to kQT queries
__global__ void __launch_bounds__(256) k_gemm_pack_queries_bf16(const float* q, int nq, int d, int dk16, uint4* qbf) {
    q += (size_t)blockIdx.y * kQT * d;       // blockIdx.y = pass
    nq -= (int)blockIdx.y * kQT;
    qbf += (size_t)blockIdx.y * (kQT / 32) * dk16 * 128;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= (kQT / 32) * dk16 * 64) return;
    const int l = o & 63;
    const int t = o >> 6;
    const int kb = t % dk16, jb = t / dk16;
    const int qi = jb * 32 + (l & 31), h = l >> 5;
    unsigned short hi[8], lo[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * kb + 8 * h + j;
        const float x = (qi < nq && k < d) ? q[(size_t)qi * d + k] : 0.f;
        split_bf16(x, hi[j], lo[j]);
    }
    uint4 vh, vl;
    __builtin_memcpy(&vh, hi, 16);
    __builtin_memcpy(&vl, lo, 16);
    qbf[(size_t)(t * 2) * 64 + l] = vh;
    qbf[(size_t)(t * 2 + 1) * 64 + l] = vl;
}

__device__ __forceinline__ uint4 ld_nt(const uint4* p) {      // non-temporal: the gallery fragments are streamed once per pass
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    const v4u v = __builtin_nontemporal_load((const v4u*)p);
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const uint4 v) {
    bf16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}

// Same contract as k_gemm_proxy (one wave: 32 rows x 64 queries, MODE 0 sample / MODE 1 append), bf16-split operands.
// Dynamic LDS: one slab of the query tile: 2 query blocks x sk16 k-blocks x {hi, lo} x 64 uint4 (128 KiB at 512 features).
template <int MODE>
__global__ void __launch_bounds__(kGemmBlock, kGemmMinBlocks) k_gemm_proxy_bf16(const uint4* __restrict__ gb, const float* __restrict__ gnorm,
                                                                    const uint4* qbf, int64_t n, int64_t row_begin,
                                                                    int64_t row_end, int dk16, const float* tau, unsigned long long* lists, int* counts, float* sample,
                                                                    int sample_rows) {
    extern __shared__ __attribute__((aligned(16))) uint4 lqb[];
    __shared__ float tau_s[kQT];
    {   // blockIdx.y = pass (64 queries each); every per-pass buffer is laid out pass-major
        const size_t ps = blockIdx.y;
        qbf += ps * (kQT / 32) * dk16 * 128;
        tau += ps * kQT;
        lists += ps * kQT * kListCap;
        counts += ps * kQT;
        sample += ps * kQT * sample_rows;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    // MODE 0 (the short sample pass) reads the query fragments straight from global memory (L2-resident) and is launched
    // WITHOUT dynamic LDS, so its small workgroups fit next to the full pass's 128-KiB workgroups on the same CUs.
    const int sk16 = MODE == 0 ? dk16 : (dk16 < kSlab16 ? dk16 : kSlab16);
    const int nslab = (dk16 + sk16 - 1) / sk16;
    const uint4* bsrc = MODE == 0 ? qbf : lqb;
    if (MODE == 1 && threadIdx.x < kQT) tau_s[threadIdx.x] = tau[threadIdx.x];
    const int64_t rb_begin = row_begin / 32, rb_end = (row_end + 31) / 32;
    const int64_t nrg = (rb_end - rb_begin + wpb - 1) / wpb;
    bool staged = MODE == 0;
    for (int64_t rg = blockIdx.x; rg < nrg; rg += gridDim.x) {
        const int64_t rb = rb_begin + rg * wpb + wave;
        const bool active = rb < rb_end;
        f32x16 acc0 = {0.f}, acc1 = {0.f};
        for (int sl = 0; sl < nslab; ++sl) {
            const int k0 = sl * sk16;
            const int kw = dk16 - k0 < sk16 ? dk16 - k0 : sk16;            // k-blocks in this slab (a multiple of 4)
            if (MODE == 1 && (nslab > 1 || !staged)) {
                __syncthreads();
                for (int i = threadIdx.x; i < 2 * kw * 128; i += blockDim.x) {
                    const int jb = i / (kw * 128), r = i - jb * kw * 128;
                    lqb[(size_t)jb * sk16 * 128 + r] = qbf[((size_t)jb * dk16 + k0) * 128 + r];
                }
                __syncthreads();
                staged = true;
            }
            if (!active) continue;
            const uint4* a = gb + ((size_t)rb * dk16 + k0) * 128 + lane;      // +0: hi fragment, +64: lo fragment of a k-block
            const int last = kw - 1;
            // EIGHT k-blocks (hi + lo = 2 KiB per wave each, 16 KiB per wave) in flight, non-temporal, re-issued right after use:
            // this kernel is HBM-bound (the six MFMAs of a k-block take 192 cycles, its 2 KiB take longer to arrive)
#define FIR_BF_LD(IDX) ld_nt(a + (size_t)((IDX) < last ? (IDX) : last) * 128)
#define FIR_BF_LDL(IDX) ld_nt(a + (size_t)((IDX) < last ? (IDX) : last) * 128 + 64)
            uint4 h0 = FIR_BF_LD(0), l0 = FIR_BF_LDL(0), h1 = FIR_BF_LD(1), l1 = FIR_BF_LDL(1), h2 = FIR_BF_LD(2), l2 = FIR_BF_LDL(2),
                  h3 = FIR_BF_LD(3), l3 = FIR_BF_LDL(3), h4 = FIR_BF_LD(4), l4 = FIR_BF_LDL(4), h5 = FIR_BF_LD(5), l5 = FIR_BF_LDL(5),
                  h6 = FIR_BF_LD(6), l6 = FIR_BF_LDL(6), h7 = FIR_BF_LD(7), l7 = FIR_BF_LDL(7);
            // query fragments one step ahead: block 0 hi/lo, block 1 hi/lo
            const uint4* bq0 = bsrc + (size_t)(MODE == 0 ? k0 : 0) * 128 + lane;
            const uint4* bq1 = bq0 + (size_t)sk16 * 128;
            uint4 b0h = bq0[0], b0l = bq0[64], b1h = bq1[0], b1l = bq1[64];
#define FIR_BF_STEP(AH, AL, KB)                                                                                  \
            {                                                                                                    \
                const int kn = (KB) + 1 < kw ? (KB) + 1 : (KB);                                                  \
                const uint4 n0h = bq0[(size_t)kn * 128], n0l = bq0[(size_t)kn * 128 + 64];                       \
                const uint4 n1h = bq1[(size_t)kn * 128], n1l = bq1[(size_t)kn * 128 + 64];                       \
                __builtin_amdgcn_sched_barrier(0);                                                               \
                const bf16x8 ah = as_bf16x8(AH), al = as_bf16x8(AL);                                             \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, as_bf16x8(b0h), acc0, 0, 0, 0);               \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, as_bf16x8(b1h), acc1, 0, 0, 0);               \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, as_bf16x8(b0l), acc0, 0, 0, 0);               \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, as_bf16x8(b1l), acc1, 0, 0, 0);               \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, as_bf16x8(b0h), acc0, 0, 0, 0);               \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, as_bf16x8(b1h), acc1, 0, 0, 0);               \
                b0h = n0h; b0l = n0l; b1h = n1h; b1l = n1l;                                                      \
            }
#define FIR_BF_NEXT(AH, AL, OFF)                \
            {                                   \
                AH = FIR_BF_LD(kb + (OFF) + 8); \
                AL = FIR_BF_LDL(kb + (OFF) + 8);\
            }
            for (int kb = 0; kb < kw; kb += 8) {
                FIR_BF_STEP(h0, l0, kb)
                FIR_BF_NEXT(h0, l0, 0)
                FIR_BF_STEP(h1, l1, kb + 1)
                FIR_BF_NEXT(h1, l1, 1)
                FIR_BF_STEP(h2, l2, kb + 2)
                FIR_BF_NEXT(h2, l2, 2)
                FIR_BF_STEP(h3, l3, kb + 3)
                FIR_BF_NEXT(h3, l3, 3)
                if (kb + 4 < kw) {      // kw is a multiple of 4, not necessarily of 8
                    FIR_BF_STEP(h4, l4, kb + 4)
                    FIR_BF_NEXT(h4, l4, 4)
                    FIR_BF_STEP(h5, l5, kb + 5)
                    FIR_BF_NEXT(h5, l5, 5)
                    FIR_BF_STEP(h6, l6, kb + 6)
                    FIR_BF_NEXT(h6, l6, 6)
                    FIR_BF_STEP(h7, l7, kb + 7)
                    FIR_BF_NEXT(h7, l7, 7)
                }
            }
#undef FIR_BF_NEXT
#undef FIR_BF_STEP
#undef FIR_BF_LD
#undef FIR_BF_LDL
        }
        if (!active) continue;
        const int64_t nrow = rb * 32 + (lane & 31);
        const float gn_lane = nrow < n ? gnorm[nrow] : 0.0f;
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            const int q = jb * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int roff = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int64_t row = rb * 32 + roff;
                const float gn = __shfl(gn_lane, roff, 64);
                if (row >= n || row < row_begin || row >= row_end) continue;
                const float dot = jb == 0 ? acc0[reg] : acc1[reg];
                const float p = gn - 2.0f * dot;
                if (MODE == 0) {
                    if (row < sample_rows) sample[(size_t)q * sample_rows + row] = p;
                } else if (p < tau_s[q]) {
                    const int slot = atomicAdd(&counts[q], 1);
                    if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(p, (uint32_t)row);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The append pass for TWO 64-query passes at once: one wave = 32 rows x 128 queries (four accumulator tiles, twelve
// MFMAs per k-block). The gallery stream -- what bounds k_gemm_proxy_bf16 (5.7 TB/s, MFMA pipe ~25 % busy) -- is read
// once per 128 queries instead of once per 64. The query fragments of 128 queries only fit LDS 256 features at a time
// (4 query blocks x 16 k-blocks x {hi, lo} x 1 KiB = 128 KiB), so every row group of a workgroup re-stages the slabs;
// the gallery fragments keep streaming through the staging barriers (a double buffer of eight k-blocks that runs on
// into the wave's next row block). What bounds this kernel is the re-staging itself -- two workgroup barriers around
// a 128 KiB copy per slab; changing the gallery ring into the double buffer left its 552 us per pass unchanged, whereas the
// fp16 kernel below, whose 128-query tile stays resident, went from 358 to 200 us. blockIdx.y = pair of passes; the per-pass scratch (tau, lists, counts, query fragments) is
// laid out pass-major, so the pair's 128 queries are simply consecutive.
// ---------------------------------------------------------------------------------------------
constexpr int kSlabW = 16;             // k-blocks (of 16 features) of the 128-query slab
constexpr int kUnitW = kSlabW / 2;     // k-blocks per unit of the gallery double buffer (hi + lo: 16 KiB per wave); dk16 is a multiple of it
constexpr int kWideLds = 4 * kSlabW * 128 * (int)sizeof(uint4);   // 128 KiB
__global__ void __launch_bounds__(kGemmBlock, 1) k_gemm_proxy_bf16_wide(const uint4* __restrict__ gb, const float* __restrict__ gnorm,
                                                                         const uint4* qbf, int64_t n, int dk16, const float* tau,
                                                                         unsigned long long* lists, int* counts) {
    extern __shared__ __attribute__((aligned(16))) uint4 lqb[];
    __shared__ float tau_s[2 * kQT];
    {
        const size_t pr = blockIdx.y;
        qbf += pr * 4 * dk16 * 128;
        tau += pr * 2 * kQT;
        lists += pr * 2 * kQT * kListCap;
        counts += pr * 2 * kQT;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    if (threadIdx.x < 2 * kQT) tau_s[threadIdx.x] = tau[threadIdx.x];
    const int64_t rb_end = (n + 31) / 32;
    const int64_t nrg = (rb_end + wpb - 1) / wpb;
    const int units = dk16 / kUnitW;
    int64_t rg = blockIdx.x;
    if (rg >= nrg) return;                           // uniform per workgroup
    // The gallery stream is a double buffer of kUnitW k-blocks (see k_gemm_proxy_f16): all loads of the next unit -- of this
    // row block or of the wave's next one -- leave before the 96 MFMAs of the current unit.
#define FIR_W_BLOCK(RG) (gb + (size_t)(((RG) * wpb + wave) < rb_end ? ((RG) * wpb + wave) : rb_end - 1) * dk16 * 128 + lane)
    const uint4* a_cur = FIR_W_BLOCK(rg);            // inactive waves stream a valid block and drop it
    uint4 ch[kUnitW], cl[kUnitW], nh[kUnitW], nl[kUnitW];
#pragma unroll
    for (int u = 0; u < kUnitW; ++u) {
        ch[u] = ld_nt(a_cur + (size_t)u * 128);
        cl[u] = ld_nt(a_cur + (size_t)u * 128 + 64);
    }
    for (; rg < nrg; rg += gridDim.x) {
        const int64_t rb = rg * wpb + wave;
        const bool active = rb < rb_end;
        const int64_t rgn = rg + gridDim.x;
        const uint4* a_nxt = FIR_W_BLOCK(rgn < nrg ? rgn : rg);
        f32x16 acc0 = {0.f}, acc1 = {0.f}, acc2 = {0.f}, acc3 = {0.f};
        for (int h = 0; h < units; ++h) {
            const uint4* src = h + 1 < units ? a_cur + (size_t)(h + 1) * kUnitW * 128 : a_nxt;
#pragma unroll
            for (int u = 0; u < kUnitW; ++u) {
                nh[u] = ld_nt(src + (size_t)u * 128);
                nl[u] = ld_nt(src + (size_t)u * 128 + 64);
            }
            if ((h & 1) == 0) {                                             // a slab is two units; 128 queries never fit whole
                const int k0 = h * kUnitW;
                const int kw = dk16 - k0 < kSlabW ? dk16 - k0 : kSlabW;
                __syncthreads();                                            // everyone is done with the previous slab
                for (int i = threadIdx.x; i < 4 * kw * 128; i += blockDim.x) {
                    const int jb = i / (kw * 128), r = i - jb * kw * 128;
                    lqb[(size_t)jb * kSlabW * 128 + r] = qbf[((size_t)jb * dk16 + k0) * 128 + r];
                }
                __syncthreads();
            }
            const uint4* bq = lqb + lane + (size_t)(h & 1) * kUnitW * 128;
#pragma unroll
            for (int u = 0; u < kUnitW; ++u) {
                const uint4* bu = bq + (size_t)u * 128;
                const bf16x8 ah = as_bf16x8(ch[u]), al = as_bf16x8(cl[u]);
#pragma unroll
                for (int jb = 0; jb < 4; ++jb) {
                    const bf16x8 bh = as_bf16x8(bu[(size_t)jb * kSlabW * 128]), bl = as_bf16x8(bu[(size_t)jb * kSlabW * 128 + 64]);
                    f32x16& acc = jb == 0 ? acc0 : jb == 1 ? acc1 : jb == 2 ? acc2 : acc3;
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < kUnitW; ++u) { ch[u] = nh[u]; cl[u] = nl[u]; }
        }
        a_cur = a_nxt;
        if (!active) continue;
        const int64_t nrow = rb * 32 + (lane & 31);
        const float gn_lane = nrow < n ? gnorm[nrow] : 0.0f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int q = jb * 32 + (lane & 31);
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int roff = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int64_t row = rb * 32 + roff;
                const float gn = __shfl(gn_lane, roff, 64);
                if (row >= n) continue;
                const float dot = jb == 0 ? acc0[reg] : jb == 1 ? acc1[reg] : jb == 2 ? acc2[reg] : acc3[reg];
                const float p = gn - 2.0f * dot;
                if (p < tau_s[q]) {
                    const int slot = atomicAdd(&counts[q], 1);
                    if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(p, (uint32_t)row);
                }
            }
        }
    }
#undef FIR_W_BLOCK
}

// ---------------------------------------------------------------------------------------------
// FIR_GEMM_F16: ONE fp16 MFMA term, 128 queries per gallery read. The gallery fragments are half the bytes of the bf16
// split (2 B per feature) and each k-block costs four MFMAs instead of twelve; the price is a proxy that is only good
// to 2^-10 |q||g| -- which the certificate simply carries in E (k_gemm_rerank): on feature vectors the gap between the
// best row and the 8th-best proxy is an order of magnitude wider than that, and a query it cannot certify goes through
// the exact scan like any other. Operands are scaled by powers of two into fp16's normal range (the gallery by one
// factor, every query by its own; undone per query column in the epilogue, exact), so rounding is relative (2^-11 per
// operand) for every element within 2^-27 of the largest; smaller ones lose at most 2^-38 of that largest value each.
// The query tile (4 blocks x 32 k-blocks x 1 KiB = 128 KiB at 512 features) stays in LDS for a whole pass when d <= 512;
// the gallery stream is double-buffered sixteen k-blocks (16 KiB per wave) at a time and runs on into the wave's NEXT
// row block, so it never drains between row groups.
// ---------------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16x8 as_f16x8(const uint4 v) {
    f16x8 r;
    __builtin_memcpy(&r, &v, 16);
    return r;
}
constexpr int kSlabH = 32;                                              // k-blocks (of 16 features) of the 128-query slab
constexpr int kHalfLds = 4 * kSlabH * 64 * (int)sizeof(uint4);          // 128 KiB
constexpr int kRing = 8;                                                // gallery k-blocks per double-buffer unit (8 KiB per wave); the fp16 dk16 is padded to it

// max |x| over the tiled gallery (padding is zero) -> out[0]; out must be zeroed first. Non-finite values poison it (NaN -> +inf).
__global__ void __launch_bounds__(256) k_gemm_absmax(const float4* __restrict__ gal4, int64_t count4, float* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count4; i += (int64_t)gridDim.x * 256) {
        const float4 g = gal4[i];
        const float a = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));
        const bool bad = !(g.x == g.x && g.y == g.y && g.z == g.z && g.w == g.w);
        m = fmaxf(m, bad ? __builtin_huge_valf() : a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(m));     // non-negative floats order like their bits
}

// tiled f32 gallery * scale (a power of two) -> gh[(rb * dk16 + kb)][lane] (uint4 = 8 fp16), round to nearest even
__global__ void __launch_bounds__(256) k_gemm_pack_gallery_f16(const float4* __restrict__ gal4, int64_t n, int dp4, int dk16, float scale,
                                                                uint4* __restrict__ gh, int kmax) {
    // kmax: features [0, kmax) of every row are packed, the rest of the fragments is zero (kmax = dp4 * 4: the whole row)
    const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (rb, kb, lane)
    const int64_t rblocks = (n + 31) / 32;
    if (o >= rblocks * dk16 * 64) return;
    const int l = (int)(o & 63);
    const int64_t t = o >> 6;
    const int kb = (int)(t % dk16);
    const int64_t rb = t / dk16;
    const int64_t row = rb * 32 + (l & 31);
    const int h = l >> 5;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * kb + 8 * h + j;
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

// Per query (one wave each): |q|^2, its own power-of-two scale qmul (largest |q_k| -> [2^13, 2^14)) and
// qinv = 1 / (qmul * gallery scale), which turns the MFMA result back into q.g. Queries past nq: all zero.
// A query whose scale would leave [2^-100, 2^100], or with a non-finite value, gets qinv = NaN: every proxy is then NaN,
// nothing is certified and the exact scan answers it.
// counts != NULL: the query's candidate count is cleared here too (instead of a memset of its own); win != NULL: ... and the adaptive
// pass's window and start value are set (k_gemm_adapt_init's two lines): two launches less in front of every super-batch.
__global__ void __launch_bounds__(64) k_gemm_qprep_f16(const float* __restrict__ q, int nq, int d, int gallery_exp, float* __restrict__ qnorm,
                                                        float* __restrict__ qmul, float* __restrict__ qinv, int qstride, int* __restrict__ counts = nullptr,
                                                        float* __restrict__ win = nullptr, unsigned int* __restrict__ t_bits = nullptr,
                                                        const float* __restrict__ gnorm_max_p = nullptr, float e_rel = 0.f, int nslot = 0,
                                                        unsigned int* __restrict__ smin_init = nullptr, int* __restrict__ fb_state = nullptr) {
    // smin_init / fb_state (the few-query path): the words three hipMemsetAsync calls used to preset -- each a launch of its own in front of a
    // 250-us call
    const int qi = blockIdx.x;
    if (threadIdx.x == 0) {
        if (smin_init) smin_init[qi] = 0xFF800000u;
        if (fb_state && qi == 0) { fb_state[0] = 0; fb_state[1] = 0; }
    }
    float s = 0.f, m = 0.f;
    bool bad = false;
    if (qi < nq)
        for (int k = threadIdx.x; k < d; k += 64) {
            const float x = q[(size_t)qi * qstride + k];
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
        if (m > 0.f) (void)frexpf(m, &ex);                 // m = f * 2^ex, f in [0.5, 1)
        int sh = m > 0.f ? 14 - ex : 0;                    // m * 2^sh in [2^13, 2^14)
        if (sh < -100 || sh > 100 || gallery_exp < -100 || gallery_exp > 100) bad = true;
        // bad: the proxies of this query are NaN and say nothing about any row -- a NaN norm keeps the certificate from holding
        const float qn = (bad && qi < nq) ? __builtin_nanf("") : s;
        qnorm[qi] = qn;
        qmul[qi] = bad ? 0.f : ldexpf(1.0f, sh);
        qinv[qi] = qi >= nq ? 0.f : bad ? __builtin_nanf("") : ldexpf(1.0f, -sh - gallery_exp);
        if (counts) counts[qi] = 0;
        if (win) {                                                   // (k_gemm_adapt_init)
            // (t_bits is the word the ranks exchange through memory-side atomics: it is only ever touched by atomics, so that no XCD's L2
            // holds a line of it that an atomic could be served from)
            // nslot > 0 (the K nearest rows, k_gemm_proxy_f16x<4, *>): eight slot words per query
            const unsigned int start = qi >= nq ? 0u : 0x7F800000u;
            if (qi >= nq) win[qi] = 0.f;
            else {
                const float w = 2.5f * e_rel * (qn + gnorm_max_p[0]);
                win[qi] = w + fabsf(w) * 1e-6f + 1e-30f;
            }
            if (nslot > 0) { for (int sl = 0; sl < 8; ++sl) atomicExch(&t_bits[(size_t)qi * 8 + sl], start); }
            else atomicExch(&t_bits[qi], start);
        }
    }
}

// queries * qmul -> qh[((pair * 4 + jb) * dk16 + kb)][lane] (uint4 = 8 fp16); blockIdx.y = pair of 64-query passes
__global__ void __launch_bounds__(256) k_gemm_pack_queries_f16(const float* q, int nq, int d, int dk16, const float* __restrict__ qmul, uint4* qh,
                                                                int qstride) {
    const int q_base = (int)blockIdx.y * 2 * kQT;
    qh += (size_t)blockIdx.y * 4 * dk16 * 64;
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 4 * dk16 * 64) return;
    const int l = o & 63;
    const int t = o >> 6;
    const int kb = t % dk16, jb = t / dk16;
    const int qi = q_base + jb * 32 + (l & 31), h = l >> 5;
    const float mul = qi < nq ? qmul[qi] : 0.f;
    f16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 16 * kb + 8 * h + j;
        const float x = (qi < nq && k < d) ? q[(size_t)qi * qstride + k] : 0.f;
        v[j] = (_Float16)(x * mul);
    }
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    qh[o] = u;
}

// One wave: 32 rows x 128 queries. MODE 0: proxies of rows [row_begin, row_end) -> sample; MODE 1: append rows below tau.
// Dynamic LDS: kHalfLds. blockIdx.y = pair of passes; all per-pass scratch is laid out pass-major (128 consecutive queries).
// dk16 is a multiple of kRing here (the fp16 fragments are padded to 256-feature units). The gallery stream is a double
// buffer of kRing k-blocks (16 KiB per wave, 128 KiB per CU in flight): the loads of the NEXT unit -- the next half of this
// row block, or the first half of the wave's next row block -- are all issued before the sixteen MFMA steps of the current
// one, so the only wait per unit is at its end, 2048 MFMA cycles after the loads left. (A ring that reloads each slot
// right after use needs counted waits inside a loop, which the compiler turns into vmcnt(0) per step: one memory latency
// per k-block -- measured 358 us per pass against 158 us of gallery stream.)
// STREAMED = 1 (rows longer than 512 features: the 128-query tile does not fit LDS): the query fragments go through two
// 64-KiB LDS buffers of one unit (16 k-blocks) each, filled by global_load_lds (no registers, asynchronous): the slab of
// the NEXT unit is requested before the current unit's MFMAs, lands under them, and one workgroup barrier per unit
// publishes it. (Staging slabs with ordinary loads between two barriers, as the resident form does once per pass, cost
// 40 % at d = 1280: 740 us per pass against 394 us of gallery stream.)
template <int MODE, int STREAMED>
__global__ void __launch_bounds__(kGemmBlock, 1) k_gemm_proxy_f16(const uint4* __restrict__ gh, const float* __restrict__ gnorm, const uint4* qh,
                                                                   const float* __restrict__ qinv, int64_t n, int64_t row_begin, int64_t row_end,
                                                                   int dk16, const float* tau, unsigned long long* lists, int* counts,
                                                                   float* sample, int sample_rows, int share, int nt) {
    extern __shared__ __attribute__((aligned(16))) uint4 lqb[];
    __shared__ float tau_s[2 * kQT], qinv_s[2 * kQT];
    // Which pair of passes (128 queries) and which row groups this workgroup takes.
    // share == 0: blockIdx.y = pair, row groups blockIdx.x, + gridDim.x, ... (the sample pass).
    // share == P (a power of two, the full pass): ONE workgroup per CU, all resident at once; the P pairs of the launch read the
    // gallery TOGETHER: the row groups are cut into gridDim.x / P contiguous ranges and P workgroups -- one per pair, placed on the
    // same XCD (blocks b and b + 8 share one) -- walk the same range at the same time, so the fp16 gallery stream leaves HBM once
    // per P * 128 queries and the other P - 1 readers hit in the XCD's L2 (or, when they drift apart, in the Infinity Cache).
    int pair_of_wg = (int)blockIdx.y;
    int64_t rg_first = blockIdx.x, rg_step = gridDim.x, rg_last = -1;      // rg_last < 0: up to nrg
    if (share > 0) {
        const int w = (int)blockIdx.x, xcd = w & 7, slot = w >> 3;
        const int ranges = ((int)gridDim.x >> 3) / share * 8;             // whole groups of `share` slots per XCD
        const int range = xcd + 8 * (slot / share);
        if (range >= ranges) return;                                       // uniform per workgroup
        pair_of_wg = slot % share;
        const int64_t nrg_all = (((row_end + 31) / 32 - row_begin / 32) + (blockDim.x >> 6) - 1) / (blockDim.x >> 6);
        rg_first = nrg_all * range / ranges;
        rg_last = nrg_all * (range + 1) / ranges;
        rg_step = 1;
    }
    {
        const size_t pr = (size_t)pair_of_wg;
        qh += pr * 4 * dk16 * 64;
        qinv += pr * 2 * kQT;
        tau += pr * 2 * kQT;
        lists += pr * 2 * kQT * kListCap;
        counts += pr * 2 * kQT;
        sample += pr * 2 * kQT * ((sample_rows + 31) / 32);      // MODE 0 writes one minimum per (row block, query)
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wpb = blockDim.x >> 6;
    if (threadIdx.x < 2 * kQT) {
        tau_s[threadIdx.x] = MODE == 1 ? tau[threadIdx.x] : 0.f;
        qinv_s[threadIdx.x] = qinv[threadIdx.x];
    }
    const int64_t rb_begin = row_begin / 32, rb_end = (row_end + 31) / 32;
    const int64_t nrg = (rb_end - rb_begin + wpb - 1) / wpb;
    const int units = dk16 / kRing;                  // units of kRing k-blocks per row block
    const int nslab = (dk16 + kSlabH - 1) / kSlabH;
    const int64_t rg_end = rg_last >= 0 ? rg_last : nrg;
    int64_t rg = rg_first;
    if (rg >= rg_end) return;                        // uniform per workgroup
    // streamed once per launch from HBM: non-temporal; shared between the pairs of a launch: ordinary loads, so that the line
    // stays in L2 for the other readers
#define FIR_H_LD(P) (nt ? ld_nt(P) : *(P))
#define FIR_H_BLOCK(RG) (gh + (size_t)((rb_begin + (RG) * wpb + wave) < rb_end ? (rb_begin + (RG) * wpb + wave) : rb_end - 1) * dk16 * 64 + lane)
    const uint4* a_cur = FIR_H_BLOCK(rg);            // waves past the last row block stream a valid one and drop the result
    uint4 cur[kRing], nxt[kRing];
#pragma unroll
    for (int u = 0; u < kRing; ++u) cur[u] = FIR_H_LD(a_cur + (size_t)u * 64);
    constexpr int kUnitsPerSlab = kSlabH / kRing;
    const bool resident = !STREAMED && nslab == 1;   // the whole 128-query tile stays in LDS for the launch (d <= 512)
    // STREAMED: unit hq of the query tile -> LDS buffer bsel; every wave moves its share of the 4 * kRing one-KiB pieces
    // (piece = (query block jb, k-block kb): 64 lanes x 16 B, contiguous in qh and in LDS)
    auto request_slab = [&](int hq, int bsel) {
        uint4* dst = lqb + (size_t)bsel * 4 * kRing * 64;
        const int per_wave = 4 * kRing / wpb;
        for (int c = 0; c < per_wave; ++c) {
            const int piece = wave * per_wave + c, jb = piece / kRing, kb = piece - jb * kRing;
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(qh + ((size_t)jb * dk16 + (size_t)hq * kRing + kb) * 64 + lane),
                                             (void __attribute__((address_space(3)))*)(dst + (size_t)(kb * 4 + jb) * 64), 16, 0, 0);
        }
    };
    int tsel = 0;                                    // STREAMED: LDS buffer of the current unit
    if (STREAMED) {
        request_slab(0, 0);
        __builtin_amdgcn_s_waitcnt(0);               // (also the prologue's gallery loads: once per kernel)
        __syncthreads();
    } else if (resident) {
        for (int i = threadIdx.x; i < 4 * dk16 * 64; i += blockDim.x) {
            const int jb = i / (dk16 * 64), r = i - jb * dk16 * 64;
            lqb[(size_t)((r >> 6) * 4 + jb) * 64 + (r & 63)] = qh[(size_t)jb * dk16 * 64 + r];
        }
        __syncthreads();
    }
    // LDS image of the query tile: k-block major, the four query blocks of a k-block side by side -- (kb * 4 + jb) * 1 KiB -- so that
    // the reads of a unit sit within the 64 KiB an LDS instruction's immediate offset reaches from ONE base register
    constexpr int jstride = 64;                                      // uint4s between the query blocks of a k-block
    // the four query fragments of the k-block the next MFMA group consumes: always read one group ahead
    uint4 b0, b1, b2, b3, c0, c1, c2, c3;
    if (resident) { const uint4* bu = lqb + lane; b0 = bu[0]; b1 = bu[jstride]; b2 = bu[2 * jstride]; b3 = bu[3 * jstride]; }
    for (; rg < rg_end; rg += rg_step) {
        const int64_t rb = rb_begin + rg * wpb + wave;
        const bool active = rb < rb_end;
        const int64_t rgn = rg + rg_step;
        const uint4* a_nxt = FIR_H_BLOCK(rgn < rg_end ? rgn : rg);
        f32x16 acc0 = {0.f}, acc1 = {0.f}, acc2 = {0.f}, acc3 = {0.f};
        float4 gn4[4];                               // this lane's 16 squared row norms (rows 8g + 4h + 0..3 of the block), for the epilogue
        const bool full_block = active && rb * 32 >= row_begin && rb * 32 + 32 <= row_end && rb * 32 + 32 <= n && (MODE == 1 || rb * 32 + 32 <= sample_rows);
        // one unit of kRing k-blocks: the loads of the NEXT unit go into N while the MFMAs consume C. Units are taken in pairs with
        // the two buffers swapping roles, so that no register copies sit between the MFMAs (a `cur = nxt` copy per unit was one
        // vector instruction per MFMA on a SIMD whose vector issue the MFMAs and the epilogue already fill to ~85 %)
        auto unit = [&](uint4 (&C)[kRing], uint4 (&N)[kRing], int h) {
            const uint4* src = h + 1 < units ? a_cur + (size_t)(h + 1) * kRing * 64 : a_nxt;
            if (STREAMED) request_slab(h + 1 < units ? h + 1 : 0, tsel ^ 1);   // its last readers passed the barrier that ended the previous unit
            if (nt) {
#pragma unroll
                for (int u = 0; u < kRing; ++u) N[u] = ld_nt(src + (size_t)u * 64);
            } else {
#pragma unroll
                for (int u = 0; u < kRing; ++u) N[u] = src[(size_t)u * 64];
            }
            if (h == units - 1 && full_block) {
                const float4* gp = (const float4*)(gnorm + rb * 32 + 4 * (lane >> 5));
#pragma unroll
                for (int g = 0; g < 4; ++g) gn4[g] = gp[2 * g];
            }
            if (!STREAMED && !resident && (h % kUnitsPerSlab) == 0) {      // forced non-streamed form with rows longer than a slab: re-stage
                const int k0 = h * kRing;
                const int kw = dk16 - k0 < kSlabH ? dk16 - k0 : kSlabH;
                __syncthreads();                                            // everyone is done with the previous slab
                for (int i = threadIdx.x; i < 4 * kw * 64; i += blockDim.x) {
                    const int jb = i / (kw * 64), r = i - jb * kw * 64;
                    lqb[(size_t)((r >> 6) * 4 + jb) * 64 + (r & 63)] = qh[((size_t)jb * dk16 + k0) * 64 + r];
                }
                __syncthreads();
            }
            const uint4* bq = STREAMED ? lqb + lane + (size_t)tsel * 4 * kRing * 64 : lqb + lane + (size_t)(h % kUnitsPerSlab) * kRing * 4 * 64;
            if (!resident) { b0 = bq[0]; b1 = bq[jstride]; b2 = bq[2 * jstride]; b3 = bq[3 * jstride]; }
            // where the fragments of the k-block AFTER this unit live, when that is known now (resident tile): the pipeline runs on
            // through the unit boundary and through the epilogue into the wave's next row block
            const uint4* bq_after = lqb + lane + (size_t)((h + 1 < units ? h + 1 : 0) % kUnitsPerSlab) * kRing * 4 * 64;
            // two sets of query fragments, (b0..b3) and (c0..c3), swap roles every k-block: the reads of k-block u + 1 go into the
            // set the MFMAs of k-block u do not use. (A rotation `b = next` through one set was compiled into eight 64-bit
            // register copies per k-block -- two vector instructions per MFMA on an already full vector issue port.)
#define FIR_H_STEP(U, X0, X1, X2, X3, Y0, Y1, Y2, Y3)                                                              \
            {                                                                                                      \
                if ((U) + 1 < kRing) {                                                                             \
                    const uint4* bu = bq + (size_t)((U) + 1) * 4 * 64;                                             \
                    Y0 = bu[0]; Y1 = bu[jstride]; Y2 = bu[2 * jstride]; Y3 = bu[3 * jstride];                      \
                } else if (resident) {                                                                             \
                    Y0 = bq_after[0]; Y1 = bq_after[jstride]; Y2 = bq_after[2 * jstride]; Y3 = bq_after[3 * jstride]; \
                }                                                                                                  \
                __builtin_amdgcn_sched_barrier(0); /* the four reads of the next group stay ahead of this group's MFMAs */ \
                const f16x8 av = as_f16x8(C[U]);                                                                   \
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, as_f16x8(X0), acc0, 0, 0, 0);                    \
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, as_f16x8(X1), acc1, 0, 0, 0);                    \
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, as_f16x8(X2), acc2, 0, 0, 0);                    \
                acc3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, as_f16x8(X3), acc3, 0, 0, 0);                    \
            }
            static_assert(kRing == 8, "the unrolled k-block steps below are written out for kRing = 8");
            FIR_H_STEP(0, b0, b1, b2, b3, c0, c1, c2, c3)
            FIR_H_STEP(1, c0, c1, c2, c3, b0, b1, b2, b3)
            FIR_H_STEP(2, b0, b1, b2, b3, c0, c1, c2, c3)
            FIR_H_STEP(3, c0, c1, c2, c3, b0, b1, b2, b3)
            FIR_H_STEP(4, b0, b1, b2, b3, c0, c1, c2, c3)
            FIR_H_STEP(5, c0, c1, c2, c3, b0, b1, b2, b3)
            FIR_H_STEP(6, b0, b1, b2, b3, c0, c1, c2, c3)
            FIR_H_STEP(7, c0, c1, c2, c3, b0, b1, b2, b3)
#undef FIR_H_STEP
            if (STREAMED) {
                __builtin_amdgcn_s_waitcnt(0);       // this wave's pieces of the next slab have landed (they left before N's loads)
                __syncthreads();                     // ... and everyone's; and nobody still reads the buffer the next request overwrites
                tsel ^= 1;
            }
        };
        int h = 0;
        for (; h + 1 < units; h += 2) {
            unit(cur, nxt, h);
            unit(nxt, cur, h + 1);
        }
        if (h < units) {                             // an odd number of units per row block: the buffers end up swapped once
            unit(cur, nxt, h);
#pragma unroll
            for (int u = 0; u < kRing; ++u) cur[u] = nxt[u];
        }
        a_cur = a_nxt;
        if (!active) continue;
        if (full_block) {
            // every row of the block is a live row of the pass: p = |g|^2 - 2 q.g per accumulator, one running minimum per lane and
            // tile; only a lane whose minimum is below its query's tau looks at its 16 values again
            const float gnv[16] = {gn4[0].x, gn4[0].y, gn4[0].z, gn4[0].w, gn4[1].x, gn4[1].y, gn4[1].z, gn4[1].w,
                                   gn4[2].x, gn4[2].y, gn4[2].z, gn4[2].w, gn4[3].x, gn4[3].y, gn4[3].z, gn4[3].w};
#pragma unroll
            for (int jb = 0; jb < 4; ++jb) {
                const int q = jb * 32 + (lane & 31);
                const float m2 = 2.0f * qinv_s[q];
                const float tq = tau_s[q];
                float pv[16];
                float mn = __builtin_huge_valf();
#pragma unroll
                for (int reg = 0; reg < 16; ++reg) {
                    const float dot = jb == 0 ? acc0[reg] : jb == 1 ? acc1[reg] : jb == 2 ? acc2[reg] : acc3[reg];
                    pv[reg] = __builtin_fmaf(-m2, dot, gnv[reg]);            // one instruction: the proxy only has to be the same number wherever it is compared
                    mn = fminf(mn, pv[reg]);                                 // NaN never enters, like k_gemm_tau's ordering
                }
                if (MODE == 1) {
                    if (mn < tq) {
#pragma unroll
                        for (int reg = 0; reg < 16; ++reg) {
                            if (pv[reg] < tq) {
                                const int64_t row = rb * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                                const int slot = atomicAdd(&counts[q], 1);
                                if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(pv[reg], (uint32_t)row);
                            }
                        }
                    }
                } else {
                    const float o = __shfl_xor(mn, 32, 64);
                    mn = o < mn ? o : mn;
                    if (lane < 32) sample[(size_t)(rb - rb_begin) * (2 * kQT) + q] = mn;
                }
            }
            continue;
        }
        // a block that straddles the end of the rows (or of the sample): row by row
        const int64_t nrow = rb * 32 + (lane & 31);
        const float gn_lane = nrow < n ? gnorm[nrow] : 0.0f;
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const int q = jb * 32 + (lane & 31);
            const float m2 = 2.0f * qinv_s[q];
            const float tq = tau_s[q];
            float mn = __builtin_huge_valf();
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) {
                const int roff = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
                const int64_t row = rb * 32 + roff;
                const float gn = __shfl(gn_lane, roff, 64);
                if (row >= n || row < row_begin || row >= row_end) continue;
                const float dot = jb == 0 ? acc0[reg] : jb == 1 ? acc1[reg] : jb == 2 ? acc2[reg] : acc3[reg];
                const float p = gn - m2 * dot;
                if (MODE == 0) {
                    if (row < sample_rows) mn = p < mn ? p : mn;         // NaN never enters, like k_gemm_tau's ordering
                } else if (p < tq) {
                    const int slot = atomicAdd(&counts[q], 1);
                    if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(p, (uint32_t)row);
                }
            }
            if (MODE == 0) {
                // what tau needs from the sample is one minimum per (row block, query): sample[(row block) * 128 + query],
                // the two half-waves hold the block's rows 4..7 mod 8 and 0..3 mod 8
                const float o = __shfl_xor(mn, 32, 64);
                mn = o < mn ? o : mn;
                if (lane < 32) sample[(size_t)(rb - rb_begin) * (2 * kQT) + q] = mn;
            }
        }
    }
#undef FIR_H_BLOCK
#undef FIR_H_LD
}

#include "fir_gemm_f16x.h"
#include "fir_gemm_regtile.h"

// tau[q] = kCand-th smallest sampled proxy, nudged up so that ties with it are appended too. One block per
// query, ONE pass over the samples: every thread keeps its kCand smallest keys sorted in registers, then kCand
// rounds of block-min pop the global order statistics.
// The bound is also kept at least one rounding window (k_gemm_rerank) above the SMALLEST sampled proxy: on a clustered
// gallery dozens of rows lie within the window of the best one, the kCand-th smallest would cut through them and the
// certificate could never hold (measured: 75 of 200 queries of a 600-identity gallery went to the exact scan).
__global__ void __launch_bounds__(256) k_gemm_tau(const float* __restrict__ sample, int sample_rows, float* __restrict__ tau,
                                                   int nq_valid, const float* __restrict__ qnorm, const float* __restrict__ gnorm_max_p, float e_rel,
                                                   int qgroup = 1, int kth_window = 0) {
    // kth_window (top-K, K <= kCand): the bound is the kCand-th smallest sampled proxy PLUS one window -- at least K rows
    // lie at or below the kCand-th smallest, and every row within a window of the K-th best one must be appended for the
    // certificate of rank K to be able to hold
    // qgroup > 1 (fp16 flow): the sample of a group of qgroup queries is laid out [value][query of the group] and holds one
    // MINIMUM per 32-row block instead of every proxy: the kCand-th smallest block minimum still has kCand sampled rows at or
    // below it, which is all the append pass needs, and is within a rank or two of the exact order statistic
    __shared__ unsigned long long red[4];
    const int q = blockIdx.x;
    if (q >= nq_valid) {        // padding queries of a half-filled pass pair: nothing is appended for them
        if (threadIdx.x == 0) tau[q] = -__builtin_huge_valf();
        return;
    }
    const float* s = sample + (size_t)(q / qgroup) * qgroup * sample_rows + (q % qgroup);
    unsigned long long best[kCand];
#pragma unroll
    for (int i = 0; i < kCand; ++i) best[i] = kKeyNone;
    for (int i = threadIdx.x; i < sample_rows; i += 256) {
        unsigned long long v = fir::key_pack(s[(size_t)i * qgroup], (uint32_t)i);
        if (v < best[kCand - 1]) {
#pragma unroll
            for (int j = 0; j < kCand; ++j) {
                const bool sw = v < best[j];
                const unsigned long long t = best[j];
                best[j] = sw ? v : t;
                v = sw ? t : v;
            }
        }
    }
    unsigned long long m = kKeyNone, smallest = kKeyNone;
    for (int r = 0; r < kCand; ++r) {
        unsigned long long c = fir::wave_min_u64(best[0]);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
        __syncthreads();
        m = red[0];
        for (int i = 1; i < 4; ++i) m = red[i] < m ? red[i] : m;
        if (r == 0) smallest = m;
        if (m == kKeyNone) break;
        if (best[0] == m) {     // keys are unique (row index): exactly one thread pops
#pragma unroll
            for (int j = 0; j + 1 < kCand; ++j) best[j] = best[j + 1];
            best[kCand - 1] = kKeyNone;
        }
    }
    if (threadIdx.x == 0) {
        // fewer than kCand sampled rows: no bound -> +inf appends everything (the list cap then decides)
        float t = __builtin_huge_valf();
        if (m != kKeyNone) {
            float v = fir::f32_from_orderable((uint32_t)(m >> 32));
            const float v1 = fir::f32_from_orderable((uint32_t)(smallest >> 32));
            const float window = 2.5f * e_rel * (qnorm[q] + gnorm_max_p[0]);       // 2 E d of the re-rank, with room
            v = kth_window ? v + window : fmaxf(v, v1 + window);                     // (top-1) a NaN window leaves v as it is
            t = v + fabsf(v) * 1e-6f + 1e-30f;
        }
        tau[q] = t;
    }
}

#include "fir_gemm_fb.h"

// Per query: every appended entry that could still be the nearest row is re-ranked with the reference's arithmetic, then
// the certificate. One wave per query. out_key[q] = exact packed key; ok[q] = 1 when the certificate holds.
// Each proxy is within E d of its row's true |g|^2 - 2 q.g, so a row whose proxy exceeds the smallest proxy p1 by more than
// 2 E d cannot beat the row that has p1: the window [p1, p1 + 2 E d] holds every possible winner -- a handful of rows on
// ordinary data, all the near-duplicates on clustered data (a fixed number of candidates, as used before, could not
// certify those and sent them to the exact scan). The window is only how candidates are CHOSEN; what proves the answer
// is the certificate at the end, which is independent of that reasoning.
__global__ void __launch_bounds__(64) k_gemm_rerank(const unsigned long long* __restrict__ lists, const int* __restrict__ counts,
                                                     const float* __restrict__ tau, const float4* __restrict__ gal4,
                                                     const float* __restrict__ queries, const float* __restrict__ qnorm,
                                                     const float* __restrict__ gnorm_max_p, int64_t n, int d, int dp4, int64_t row_offset, float e_rel,
                                                     int ngroup, unsigned long long* __restrict__ out_key, int* __restrict__ ok, int qstride, const float4* __restrict__ rowmajor,
                                                     const RerankFb fb) {
    // q: the query's scratch slot (lists, counts, tau, qnorm); qo: its place in queries / out_key / ok -- the same for a super-batch's
    // own re-rank, the query's index in the call for a second-chance round (fir_gemm_fb.h)
    const int q = blockIdx.x, lane = threadIdx.x;
    int qo = q;
    if (fb.qmap) {
        if (q >= fb.state[0] - fb.live_off) return;               // (uniform per workgroup)
        qo = fb.qmap[q];
    }
    const int cnt = counts[q];
    const int have = cnt < kListCap ? cnt : kListCap;
    const unsigned long long* L = lists + (size_t)q * kListCap;
    unsigned long long kmin = kKeyNone;
    for (int i = lane; i < have; i += 64) {
        const unsigned long long v = L[i];
        kmin = v < kmin ? v : kmin;
    }
    kmin = fir::wave_min_u64(kmin);
    const float qn = qnorm[q], gmax = gnorm_max_p[0];
    // E bounds every rounding on both sides, in distance units: the f32 fma chain of the dot product (<= d u |q||g|,
    // doubled), the two float norms (<= d u each), the reference's own d+3 roundings of the (q-g)^2 sum and its divide:
    // (4d + 11) u (|q|^2 + |g|^2) / d in total; 8 d u (...) / d is used.  u = 2^-24.
    // e_rel = 8 d u, plus 2^-14 for the bf16-split variant (dropped lo.lo / residual terms, 3 * 2^-18 |g||q|, doubled in p)
    // or 2^-10 (1 + 2^-4) for the single fp16 term (both operands rounded to 11 bits)
    const float E = e_rel * (qn + gmax) / (float)d;
    const float p1 = kmin != kKeyNone ? fir::f32_from_orderable((uint32_t)(kmin >> 32)) : __builtin_huge_valf();
    float win = p1 + 2.0f * E * (float)d;
    win += fabsf(win) * 1e-6f;
    const float* qv = queries + (size_t)qo * qstride;
    const int d4 = (d + 3) >> 2;                                  // float4 chunks of the compared features (a prefix of the row when d < its length)
    unsigned long long best = kKeyNone;
    float p_out = __builtin_huge_valf();            // smallest proxy NOT re-ranked
    int reranked = 0;
    extern __shared__ __attribute__((aligned(16))) float4 crow[];     // [ngroup][dp4]: candidate rows, loaded by the whole wave
    const int cs = dp4 | 1;                                       // candidate-row stride in LDS, odd: the lanes' rows start in different banks
    float4* qrow = crow + (size_t)ngroup * cs;                 // [dp4]: the query, zero-padded like the gallery rows (a (0-0)^2 term adds +0)
    for (int k = lane; k < dp4 * 4; k += 64) ((float*)qrow)[k] = k < d ? qv[k] : 0.0f;
    __syncthreads();
    for (int base = 0; base < have; base += 64) {
        const int i = base + lane;
        const unsigned long long v = i < have ? L[i] : kKeyNone;
        const float p = fir::f32_from_orderable((uint32_t)(v >> 32));
        const bool in = i < have && p <= win;
        if (i < have && !in) p_out = fminf(p_out, p);
        unsigned long long mask = __ballot(in);
        reranked += __popcll(mask);
        while (mask) {                                                  // wave-uniform
            // up to ngroup (<= kRerankGroup) candidates at a time: all lanes fetch their rows into LDS, then lane g re-computes
            // candidate g's distance from there: db_features.cpp:22-42 order, un-fused (fir::accum<kL2>)
            unsigned long long mine = kKeyNone;
            int ng = 0;
            for (int g = 0; g < ngroup; ++g) {
                if (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const unsigned long long cv = __shfl((unsigned long long)v, src, 64);
                    const int64_t row = (int64_t)(uint32_t)(cv & 0xFFFFFFFFull);
                    if (rowmajor) {                                     // the row-major shadow copy: one contiguous row, coalesced
                        const float4* gr = rowmajor + (size_t)row * d4;
                        for (int c = lane; c < d4; c += 64) crow[(size_t)g * cs + c] = gr[c];
                    } else {                                            // the tiled gallery: 16 bytes per KiB
                        const float4* gr = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
                        for (int c = lane; c < d4; c += 64) crow[(size_t)g * cs + c] = gr[(size_t)c * 64];
                    }
                    if (lane == g) mine = cv;
                    ++ng;
                }
            }
            __syncthreads();
            if (lane < ng) {
                const float4* my = crow + (size_t)lane * cs;
                float acc = 0.0f;
                for (int c = 0; c < d4; ++c) {
                    const float4 g4 = my[c], q4 = qrow[c];
                    acc = fir::accum<fir::kL2>(acc, q4.x, g4.x);
                    acc = fir::accum<fir::kL2>(acc, q4.y, g4.y);
                    acc = fir::accum<fir::kL2>(acc, q4.z, g4.z);
                    acc = fir::accum<fir::kL2>(acc, q4.w, g4.w);
                }
                const float dist = acc / (float)d;
                if (dist < fir::kNotFound) {
                    const int64_t row = (int64_t)(uint32_t)(mine & 0xFFFFFFFFull);
                    const unsigned long long key = fir::key_pack(dist, (uint32_t)(row + row_offset));
                    best = key < best ? key : best;
                }
            }
            __syncthreads();
        }
    }
    best = fir::wave_min_u64(best);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p_out = fminf(p_out, __shfl_xor(p_out, off, 64));
    if (lane == 0) {
        out_key[qo] = best;
        // Certificate. Every row NOT re-ranked has a proxy >= p_excl: the smallest list entry outside the window, or tau
        // for rows that were never appended. Its reference distance is then >= (|q|^2 + p_excl)/d - E.
        bool certified = false;
        if (cnt <= kListCap) {                      // nothing below tau was dropped from the list
            const float t = tau[q];
            const float p_excl = t != t ? t : fminf(p_out, t);      // a NaN bound must not certify anything
            const float lower = (qn + p_excl) / (float)d - E;
            const float bd = best != kKeyNone ? fir::f32_from_orderable((uint32_t)(best >> 32)) : fir::kNotFound;
            certified = lower > bd;                 // false for NaN
            if (n <= reranked) certified = true;    // every row was re-ranked
        }
        ok[qo] = certified ? 1 : 0;
        if (!certified && fb.list && !fb.qmap) {
            // a second pass may append below min(this pass's bound, smallest stored proxy + one window): at or above the smallest proxy of
            // ALL rows + the window (a stored proxy is some row's), so that list will hold every possible winner. NaN bounds stay NaN
            // (nothing is appended, nothing certified: the exact scan answers).
            const float w = 2.5f * e_rel * (qn + gmax);
            const float c = p1 + (w + fabsf(w) * 1e-6f + 1e-30f);
            const float t = tau[q];
            fb.tau2[fb.q_base + q] = c < t ? c : t;
            fb.list[atomicAdd(&fb.state[0], 1)] = fb.q_base + q;
            fb_note(fb.state, cnt, t, p1, qn);
        }
    }
}

// ---- calls of 1..8 queries against a gallery that does not fit the caches ----
// Such a call is one gallery pass whatever it computes, so it is bound by the bytes of that pass: the exact f32 scan reads n*d*4
// (337 us at 1M x 512), this nomination scan reads the fp16 copy, n*d*2. No matrix cores (eight queries would fill 1/16 of a
// tile): v_dot2c_f32_f16 on the same fragments, lane l = row l & 31, k-half l >> 5. Every proxy is written out (n floats per
// query) together with the smallest one per query; tau = smallest proxy of ALL rows + one window (k_gemm_tau_min), the rows
// below it are collected (k_gemm_select) and re-ranked exactly with the usual certificate (k_gemm_rerank) -- which holds by
// construction here: every excluded row is a full window above the smallest proxy.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
template <int NQ>
__global__ void __launch_bounds__(256) k_gemm_scan_f16(const uint4* __restrict__ gh, const float* __restrict__ gnorm, const uint4* __restrict__ qh,
                                                        const float* __restrict__ qinv, int64_t n, int dk16, float* __restrict__ proxies,
                                                        unsigned int* __restrict__ smin) {
    extern __shared__ __attribute__((aligned(16))) uint4 qsf[];           // [k-block][k-half][query]
    for (int idx = threadIdx.x; idx < dk16 * 2 * NQ; idx += blockDim.x) {
        const int i = idx % NQ, h = (idx / NQ) & 1, kb = idx / (2 * NQ);
        qsf[idx] = qh[(size_t)kb * 64 + 32 * h + i];                      // queries 0..7 sit in query block 0 of the pair's fragments
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, h = lane >> 5;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nw = (int64_t)gridDim.x * (blockDim.x >> 6);
    const int64_t nrb = (n + 31) / 32;
    float m2[NQ], smallest[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) { m2[i] = 2.0f * qinv[i]; smallest[i] = __builtin_huge_valf(); }
    for (int64_t rb = gw; rb < nrb; rb += nw) {
        const uint4* a = gh + (size_t)rb * dk16 * 64 + lane;
        float acc[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) acc[i] = 0.f;
        for (int kb0 = 0; kb0 < dk16; kb0 += 8) {                          // dk16 is a multiple of 8
            uint4 g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = ld_nt(a + (size_t)(kb0 + u) * 64);
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                f16x2 gv[4];
                __builtin_memcpy(gv, &g[u], 16);
#pragma unroll
                for (int i = 0; i < NQ; ++i) {
                    const uint4 qq = qsf[((kb0 + u) * 2 + h) * NQ + i];
                    f16x2 qv[4];
                    __builtin_memcpy(qv, &qq, 16);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i] = __builtin_amdgcn_fdot2(gv[t], qv[t], acc[i], false);
                }
            }
        }
        const int64_t row = rb * 32 + (lane & 31);
        const float gn = row < n ? gnorm[row] : 0.f;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float dot = acc[i] + __shfl_xor(acc[i], 32, 64);        // the two k-halves of the row
            if (lane < 32 && row < n) {
                const float p = __builtin_fmaf(-m2[i], dot, gn);
                proxies[(size_t)i * n + row] = p;
                smallest[i] = fminf(smallest[i], p);                       // NaN never enters
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        float v = smallest[i];
#pragma unroll
        for (int off = 16; off >= 1; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
        if (lane == 0 && v < __builtin_huge_valf()) atomicMin(&smin[i], fir::f32_orderable(v));
    }
}

// lists[q] <- every row whose proxy is below tau[q] (blockIdx.y = query)
__global__ void __launch_bounds__(256) k_gemm_select(const float* __restrict__ proxies, int64_t n, const float* __restrict__ tau,
                                                      unsigned long long* __restrict__ lists, int* __restrict__ counts) {
    const int q = blockIdx.y;
    const float tq = tau[q];
    const float* p = proxies + (size_t)q * n;
    for (int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x; row < n; row += (int64_t)gridDim.x * 256) {
        const float v = p[row];
        if (v < tq) {
            const int slot = atomicAdd(&counts[q], 1);
            if (slot < kListCap) lists[(size_t)q * kListCap + slot] = fir::key_pack(v, (uint32_t)row);
        }
    }
}

// The K nearest rows (K <= kTopKMax) from the same lists: as k_gemm_rerank with the window hung on the K-th smallest proxy
// of the list, every lane keeping the K smallest exact keys it computed, K rounds of wave-minimum at the end, and the
// certificate taken against the K-th exact distance: no row outside the re-ranked set can be among the K nearest, nor tie
// with the K-th (strict '>'). out_key[q * k + r], ascending; kKeyNone where fewer than K rows are below 100000.
constexpr int kTopKMax = 8;
__global__ void __launch_bounds__(64) k_gemm_rerank_topk(const unsigned long long* __restrict__ lists, const int* __restrict__ counts,
                                                          const float* __restrict__ tau, const float4* __restrict__ gal4,
                                                          const float* __restrict__ queries, const float* __restrict__ qnorm,
                                                          const float* __restrict__ gnorm_max_p, int64_t n, int d, int dp4, int64_t row_offset,
                                                          float e_rel, int ngroup, int k, unsigned long long* __restrict__ out_key,
                                                          int* __restrict__ ok, int qstride, const float4* __restrict__ rowmajor, const RerankFb fb) {
    const int q = blockIdx.x, lane = threadIdx.x;                 // (q, qo: as k_gemm_rerank)
    int qo = q;
    if (fb.qmap) {
        if (q >= fb.state[0] - fb.live_off) return;
        qo = fb.qmap[q];
    }
    const int cnt = counts[q];
    const int have = cnt < kListCap ? cnt : kListCap;
    const unsigned long long* L = lists + (size_t)q * kListCap;
    // the K-th smallest list entry (keys are unique: the row is part of the key)
    unsigned long long prev = 0, kth = kKeyNone;
    int found = 0;
    for (int r = 0; r < k; ++r) {
        unsigned long long m = kKeyNone;
        for (int i = lane; i < have; i += 64) {
            const unsigned long long v = L[i];
            if ((r == 0 || v > prev) && v < m) m = v;
        }
        m = fir::wave_min_u64(m);
        if (m == kKeyNone) break;
        prev = m;
        kth = m;
        ++found;
    }
    const float qn = qnorm[q], gmax = gnorm_max_p[0];
    const float E = e_rel * (qn + gmax) / (float)d;                   // see k_gemm_rerank
    const float pk = found == k ? fir::f32_from_orderable((uint32_t)(kth >> 32)) : __builtin_huge_valf();   // a short list is re-ranked whole
    float win = pk + 2.0f * E * (float)d;
    win += fabsf(win) * 1e-6f;
    const float* qv = queries + (size_t)qo * qstride;
    const int d4 = (d + 3) >> 2;                                  // float4 chunks of the compared features (a prefix of the row when d < its length)
    unsigned long long best[kTopKMax];
#pragma unroll
    for (int i = 0; i < kTopKMax; ++i) best[i] = kKeyNone;
    float p_out = __builtin_huge_valf();            // smallest proxy NOT re-ranked
    int reranked = 0;
    extern __shared__ __attribute__((aligned(16))) float4 crow[];     // [ngroup][dp4] candidate rows, then [dp4] the query
    const int cs = dp4 | 1;                                       // candidate-row stride in LDS, odd: the lanes' rows start in different banks
    float4* qrow = crow + (size_t)ngroup * cs;
    for (int c = lane; c < dp4 * 4; c += 64) ((float*)qrow)[c] = c < d ? qv[c] : 0.0f;
    __syncthreads();
    for (int base = 0; base < have; base += 64) {
        const int i = base + lane;
        const unsigned long long v = i < have ? L[i] : kKeyNone;
        const float p = fir::f32_from_orderable((uint32_t)(v >> 32));
        const bool in = i < have && p <= win;
        if (i < have && !in) p_out = fminf(p_out, p);
        unsigned long long mask = __ballot(in);
        reranked += __popcll(mask);
        while (mask) {                                                  // wave-uniform
            unsigned long long mine = kKeyNone;
            int ng = 0;
            for (int g = 0; g < ngroup; ++g) {
                if (mask) {
                    const int src = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const unsigned long long cv = __shfl((unsigned long long)v, src, 64);
                    const int64_t row = (int64_t)(uint32_t)(cv & 0xFFFFFFFFull);
                    if (rowmajor) {                                     // the row-major shadow copy: one contiguous row, coalesced
                        const float4* gr = rowmajor + (size_t)row * d4;
                        for (int c = lane; c < d4; c += 64) crow[(size_t)g * cs + c] = gr[c];
                    } else {                                            // the tiled gallery: 16 bytes per KiB
                        const float4* gr = gal4 + (size_t)(row >> 6) * dp4 * 64 + (row & 63);
                        for (int c = lane; c < d4; c += 64) crow[(size_t)g * cs + c] = gr[(size_t)c * 64];
                    }
                    if (lane == g) mine = cv;
                    ++ng;
                }
            }
            __syncthreads();
            if (lane < ng) {
                const float4* my = crow + (size_t)lane * cs;
                float acc = 0.0f;
                for (int c = 0; c < d4; ++c) {
                    const float4 g4 = my[c], q4 = qrow[c];
                    acc = fir::accum<fir::kL2>(acc, q4.x, g4.x);
                    acc = fir::accum<fir::kL2>(acc, q4.y, g4.y);
                    acc = fir::accum<fir::kL2>(acc, q4.z, g4.z);
                    acc = fir::accum<fir::kL2>(acc, q4.w, g4.w);
                }
                const float dist = acc / (float)d;
                if (dist < fir::kNotFound) {
                    const int64_t row = (int64_t)(uint32_t)(mine & 0xFFFFFFFFull);
                    unsigned long long key = fir::key_pack(dist, (uint32_t)(row + row_offset));
#pragma unroll
                    for (int j = 0; j < kTopKMax; ++j) {                // sorted insert
                        const bool sw = key < best[j];
                        const unsigned long long t = best[j];
                        best[j] = sw ? key : t;
                        key = sw ? t : key;
                    }
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p_out = fminf(p_out, __shfl_xor(p_out, off, 64));
    unsigned long long kth_exact = kKeyNone;
    for (int r = 0; r < k; ++r) {
        const unsigned long long m = fir::wave_min_u64(best[0]);
        if (lane == 0) out_key[(size_t)qo * k + r] = m;
        kth_exact = m;
        if (m != kKeyNone && best[0] == m) {                            // exactly one lane holds it
#pragma unroll
            for (int j = 0; j + 1 < kTopKMax; ++j) best[j] = best[j + 1];
            best[kTopKMax - 1] = kKeyNone;
        }
    }
    if (lane == 0) {
        bool certified = false;
        if (cnt <= kListCap) {                      // nothing below tau was dropped from the list
            const float t = tau[q];
            const float p_excl = t != t ? t : fminf(p_out, t);      // a NaN bound must not certify anything
            const float lower = (qn + p_excl) / (float)d - E;
            // fewer than K rows found so far: any row below 100000 outside the re-ranked set would belong to the answer
            const float bd = kth_exact != kKeyNone ? fir::f32_from_orderable((uint32_t)(kth_exact >> 32)) : fir::kNotFound;
            certified = lower > bd;                 // false for NaN
            if (n <= reranked) certified = true;    // every row was re-ranked
        }
        ok[qo] = certified ? 1 : 0;
        if (!certified && fb.list && !fb.qmap) {
            // (as k_gemm_rerank, hung on the K-th smallest stored proxy: the K-th smallest proxy of ALL rows is at or below it)
            const float w = 2.5f * e_rel * (qn + gmax);
            const float c = pk + (w + fabsf(w) * 1e-6f + 1e-30f);
            const float t = tau[q];
            fb.tau2[fb.q_base + q] = c < t ? c : t;
            fb.list[atomicAdd(&fb.state[0], 1)] = fb.q_base + q;
            fb_note(fb.state, cnt, t, pk, qn);
        }
    }
}

// rowmajor[row * d4 + c] = chunk c of row `row`: the re-rank's copy of the compared features. A row of the tiled gallery is d4
// separate 16-byte pieces 1 KiB apart (~4x its size in 64-byte sectors per gather); here it is one contiguous run.
__global__ void __launch_bounds__(256) k_gemm_untile(const float4* __restrict__ gal4, int64_t n, int dp4, int d4, float4* __restrict__ rowmajor) {
    // one workgroup per (tile of 64 rows, group of 32 chunks): read coalesced along the rows, written coalesced along the chunks
    __shared__ float4 t[32][65];
    const int64_t tile = blockIdx.x;
    const int c0 = blockIdx.y * 32;
    for (int i = threadIdx.x; i < 32 * 64; i += 256) {
        const int c = i >> 6, r = i & 63;
        if (c0 + c < d4) t[c][r] = gal4[((size_t)tile * dp4 + c0 + c) * 64 + r];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += 256) {
        const int r = i >> 5, c = i & 31;
        const int64_t row = tile * 64 + r;
        if (row < n && c0 + c < d4) rowmajor[(size_t)row * d4 + c0 + c] = t[c][r];
    }
}

// gnorm[row] = |g|^2 (one thread per row; any summation order is covered by the certificate's E).
__global__ void __launch_bounds__(256) k_gemm_row_norms(const float4* __restrict__ gal4, int64_t n, int dp4, float* __restrict__ gnorm, int d4) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    float s = 0.f;
    for (int c = 0; c < d4; ++c) {
        const float4 g = gal4[((row >> 6) * dp4 + c) * 64 + (row & 63)];
        s += g.x * g.x + g.y * g.y + g.z * g.z + g.w * g.w;
    }
    gnorm[row] = s;
}

__global__ void k_gemm_max(const float* __restrict__ gnorm, int64_t n, float* __restrict__ out) {
    __shared__ float red[4];
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < n; i += 256) m = fmaxf(m, gnorm[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) out[0] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// The adaptive flow (k_gemm_proxy_f16x<3, *>): per query the rounding window (2.5 E d, as k_gemm_tau_min) and the start value of
// T -- +inf, or 0 for the padding queries of a half-filled pair (nothing is ever below it) ...
__global__ void __launch_bounds__(256) k_gemm_adapt_init(float* __restrict__ win, unsigned int* __restrict__ t_bits, int nq_total, int nq_valid,
                                                          const float* __restrict__ qnorm, const float* __restrict__ gnorm_max_p, float e_rel) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_total) return;
    if (q >= nq_valid) { win[q] = 0.f; t_bits[q] = 0u; return; }
    const float w = 2.5f * e_rel * (qnorm[q] + gnorm_max_p[0]);
    win[q] = w + fabsf(w) * 1e-6f + 1e-30f;                   // (a NaN window: T never falls, nothing is certified)
    t_bits[q] = 0x7F800000u;
}
// ... and, after the pass, the bound the re-rank's certificate gets: every row that was not appended has a proxy >= tau
__global__ void __launch_bounds__(256) k_gemm_adapt_final(unsigned int* __restrict__ t_bits, const float* __restrict__ qnorm, float* __restrict__ tau,
                                                           int nq_total, int nq_valid, int nslot = 0) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_total) return;
    // (read by an atomic, like every other access to these words)
    unsigned int t;
    if (nslot > 0) {                                  // the K nearest rows: the K-th smallest of the eight slot values (K = nslot)
        unsigned int v[8];
        for (int sl = 0; sl < 8; ++sl) v[sl] = atomicMin(&t_bits[(size_t)q * 8 + sl], 0xFFFFFFFFu);
        t = 0xFFFFFFFFu;
        for (int i = 0; i < 8; ++i) {
            int rank = 0;
            for (int j = 0; j < 8; ++j) rank += (v[j] < v[i] || (v[j] == v[i] && j < i)) ? 1 : 0;
            if (rank == nslot - 1) t = v[i];
        }
    } else t = atomicMin(&t_bits[q], 0xFFFFFFFFu);
    tau[q] = q < nq_valid ? __uint_as_float(t) - qnorm[q] : -__builtin_huge_valf();      // the same subtraction the kernel compares against
}

}  // namespace

struct fir_gemm {
    fir_gallery* g = nullptr;
    fir_gallery_view v;
    const float4* gal4 = nullptr;
    int dp4 = 0, dq8 = 0;
    int feat = 0;               // features compared: the gallery's d, or a prefix [0, feat) of every row (fir_gemm_create_range); v.d stays the row length
    int precision = 0;          // 0: f32 MFMA, 1: bf16 split (hi.hi + hi.lo + lo.hi), 2: one fp16 term
    uint4* gh = nullptr;        // fp16 fragments (precision 2)
    float* proxies = nullptr;   // few-query calls: every row's proxy, proxies_nq x n floats (grown on first use)
    int proxies_nq = 0;
    int lists_cap = 0;          // queries the candidate lists / counts of one scratch set are sized for (grown to the super-batch in use)
    float4* rowmajor = nullptr; // row-major f32 copy of the compared features for the re-rank's gathers (absent when HBM is short: the tiled gallery serves)
    int gallery_exp = 0;        // fp16: the gallery was multiplied by 2^gallery_exp
    float* qmul[2] = {nullptr, nullptr};
    float* qinv[2] = {nullptr, nullptr};
    int dk16 = 0;               // bf16 variant: k-blocks of 16 features (padded to a multiple of 4)
    float4* gm = nullptr;       // f32 fragments
    uint4* gb = nullptr;        // bf16 hi/lo fragments
    uint4* qbf[2] = {nullptr, nullptr};
    float* gnorm = nullptr;
    float* gmax = nullptr;
    // two sets of per-pass scratch: the re-rank of pass i runs on a side stream while pass i+1 is computed
    float4* qm[2] = {nullptr, nullptr};
    float* qnorm[2] = {nullptr, nullptr};
    float* tau[2] = {nullptr, nullptr};
    float* sample = nullptr;
    unsigned long long* lists[2] = {nullptr, nullptr};
    int* counts[2] = {nullptr, nullptr};
    hipStream_t side = nullptr;
    hipEvent_t main_done[2] = {nullptr, nullptr}, rerank_done[2] = {nullptr, nullptr}, prep_done[2] = {nullptr, nullptr};
    hipEvent_t queries_ready = nullptr;
    hipStream_t copy = nullptr;           // host-pointer calls: the queries of super-batch i + 1 are uploaded under super-batch i's full passes
    hipEvent_t copy_done[2] = {nullptr, nullptr};
    int* ok = nullptr; size_t ok_cap = 0;  // certificate flags of one call
    int sample_rows = 0;
    // uncertified queries, handled on the device in stream order (fir_gemm_fb.h)
    int* fb_state = nullptr;              // int[kFbStateWords]: count, count2, notes taken, -, running totals (2 x 64 bit), the first uncertified queries' numbers (fir_gemm_fb.h)
    int* fb_list = nullptr;               // [ok_cap] queries the first certificate did not hold for
    int* fb_list2 = nullptr;              // [ok_cap] ... still uncertified after the second-chance rounds
    float* fb_tau2 = nullptr;             // [ok_cap] the bound a second pass may append below
    float* sc_qnorm = nullptr; float* sc_qmul = nullptr; float* sc_qinv = nullptr; float* sc_tau = nullptr;   // one pair's scratch of a second-chance round
    int* sc_counts = nullptr;
    unsigned long long* sc_lists = nullptr;
    uint4* sc_qbf = nullptr;
    size_t fb_lds = 0;                    // dynamic LDS of k_gemm_exact_fb
    int fb_grid = 0;
    int64_t passes = 0;
    int rerank_group = kRerankGroup;      // candidate rows the re-rank stages in LDS at a time
    int streamed = -1;                    // fp16: query slabs through the LDS double buffer (-1: when the tile does not fit, d > 512)
    bool wide = true;                     // bf16: pairs of passes through k_gemm_proxy_bf16_wide (FIR_GEMM_WIDE=0 turns it off)
    unsigned int* smin[2] = {nullptr, nullptr};   // register-tile flow: smallest sampled proxy per query (orderable bits)
    int rt_sample_rows = 0;                // ... over this many rows (n / 32)
    int regtile = -1;                     // fp16 full pass through the register-tile kernel: -1 = where it measured faster (rows up to 256 features), 0 / 1 = never / wherever it exists (FIR_GEMM_REGTILE)                  // fp16 full pass: query fragments in registers, gallery through the LDS-DMA ring (FIR_GEMM_REGTILE=0: the LDS-tile kernel)
    int mfma16 = 0;                       // fp16: both operands in the 16-row fragment order, every pass on v_mfma_f32_16x16x32_f16 (fir_gemm_f16x.h): the default since it was
                                          // measured against the 32x32x16 kernels at the same wave tile (profiles/r03_mfma_shape_ab.txt); FIR_GEMM_MFMA16=0 brings those back
    bool f64 = false;                     // the nomination state of a float64 training set owned by fir_cls.hip (fir_gemm_f64.h)
    const double2* gal2 = nullptr; int dp2 = 0;
    int adaptive_topk = 1;                // ... and for the K nearest rows (k_gemm_proxy_f16x<4, *>: K slot minima per query); FIR_GEMM_ADAPTIVE_TOPK=0: the sample flow
    int adaptive = 1;                     // 16-row top-1 flow: the append threshold is found during the full pass (k_gemm_proxy_f16x<3, *>), no sample pass
                                          // (profiles/r03_adaptive_threshold.txt); FIR_GEMM_ADAPTIVE=0: the sample flow
    float* awin[2] = {nullptr, nullptr};  // ... its per-query windows
    unsigned int* aT[2] = {nullptr, nullptr};   // ... and the ranks' shared T (float bits)
    float erel_scale = 1.0f;              // AUDIT KNOB, never set in production: the certificate's relative error bound is multiplied by this (FIR_GEMM_EREL_SCALE);
                                          // tests/test_gpu_gemm.py shows that a bound shrunk to a quarter returns a wrong row on a crafted near-tie, i.e. that the suite can see an unsound bound
    int dbg_block = kGemmBlock;           // timing experiments (audit build, FIR_GEMM_DBG_BLOCK=256): the resident adaptive pass with four waves per workgroup = ONE wave per SIMD
    int dbg_skip = 0;                     // timing experiments: bit 0 = no epilogue, bit 1 = no gallery stream, bit 2 = no query-fragment re-reads (FIR_GEMM_DBG_SKIP; wrong answers)
    int no_block_bound = 0;               // A/B: the append forms compute all eight proxies of every query block (FIR_GEMM_NO_BLOCK_BOUND)
    int prio = 0;                         // mfma16 experiment: s_setprio 2 around the MFMA phase of a row block (FIR_GEMM_PRIO)
    int stagger = 0;                      // mfma16 experiment: the second wave of every SIMD starts half a unit late (FIR_GEMM_STAGGER)
    int share_streamed = 8;               // ... of them when the query slabs are streamed (FIR_GEMM_SHARE_STREAMED)
    int few_blocks = 1;                   // calls of <= 32 queries multiply against the live query blocks only (FIR_GEMM_FEW_BLOCKS=0: the whole tile, for A/B runs)
    int share_max = 16;                   // fp16: up to this many pairs of passes (x 128 queries) read the gallery together in one launch (FIR_GEMM_SHARE; 0 = the old one-pair-at-a-time grid)
};



// the 16-row kernels by (mode, query slabs streamed, odd number of units per row block)
typedef void (*fir_x_fn)(const uint4*, const float*, const uint4*, const float*, int64_t, int64_t, int64_t, int, const float*, unsigned long long*, int*, float*,
                         int, int, int, int, unsigned int*, int);
static fir_x_fn pick_x(int mode, bool streamed, bool odd, int dbg = 0, int njb = 8) {
    if (njb < 8 && mode == 3 && !odd && !dbg) {        // a call of <= 16 / <= 32 queries (top-1, threshold found on the way)
        if (njb == 1) return streamed ? k_gemm_proxy_f16x<3, 1, 0, 0, 1> : k_gemm_proxy_f16x<3, 0, 0, 0, 1>;
        if (njb == 2) return streamed ? k_gemm_proxy_f16x<3, 1, 0, 0, 2> : k_gemm_proxy_f16x<3, 0, 0, 0, 2>;
    }
#ifdef FIR_AUDIT      // (the timing forms are instantiated in the audit build only: the shipped library cannot select them)
    if (dbg && mode == 3 && !streamed && !odd) {       // timing experiments (FIR_GEMM_DBG_SKIP): wrong answers
        switch (dbg & 1023) {
            case 1: return k_gemm_proxy_f16x<3, 0, 0, 1>;
            case 2: return k_gemm_proxy_f16x<3, 0, 0, 2>;
            case 3: return k_gemm_proxy_f16x<3, 0, 0, 3>;
            case 5: return k_gemm_proxy_f16x<3, 0, 0, 5>;
            case 7: return k_gemm_proxy_f16x<3, 0, 0, 7>;
            case 512: return k_gemm_proxy_f16x<3, 0, 0, 512>;   // (not a wrong-answer form: the L2 touch two units ahead by one workgroup in sixteen)
            case 256: return k_gemm_proxy_f16x<3, 0, 0, 256>;   // (not a wrong-answer form: timestamps of workgroup 0's load bursts, FIR_GEMM_DUMP_PHASES)
            case 64: return k_gemm_proxy_f16x<3, 0, 0, 64>;     // (not a wrong-answer form: the gallery pieces one behind each MFMA pair of a unit's first step)
            case 33: return k_gemm_proxy_f16x<3, 0, 0, 33>;     // no MFMAs, no epilogue: the gallery stream + the query-fragment re-reads
            case 37: return k_gemm_proxy_f16x<3, 0, 0, 37>;     // ... the gallery stream alone
            case 35: return k_gemm_proxy_f16x<3, 0, 0, 35>;     // ... the query-fragment re-reads alone
            case 16: return k_gemm_proxy_f16x<3, 0, 0, 16>;     // (not a wrong-answer form either: the gallery pieces of a unit requested two per step instead of in one burst)
            case 8: return k_gemm_proxy_f16x<3, 0, 0, 8>;       // (not a wrong-answer form: the epilogue behind its own row block, for A/B runs)
            default: break;
        }
    }
#endif
    if (mode == 1) return streamed ? (odd ? k_gemm_proxy_f16x<1, 1, 1> : k_gemm_proxy_f16x<1, 1, 0>) : (odd ? k_gemm_proxy_f16x<1, 0, 1> : k_gemm_proxy_f16x<1, 0, 0>);
    if (mode == 3) return streamed ? (odd ? k_gemm_proxy_f16x<3, 1, 1> : k_gemm_proxy_f16x<3, 1, 0>) : (odd ? k_gemm_proxy_f16x<3, 0, 1> : k_gemm_proxy_f16x<3, 0, 0>);
    if (mode == 4) return streamed ? (odd ? k_gemm_proxy_f16x<4, 1, 1> : k_gemm_proxy_f16x<4, 1, 0>) : (odd ? k_gemm_proxy_f16x<4, 0, 1> : k_gemm_proxy_f16x<4, 0, 0>);
    return streamed ? (odd ? k_gemm_proxy_f16x<2, 1, 1> : k_gemm_proxy_f16x<2, 1, 0>) : (odd ? k_gemm_proxy_f16x<2, 0, 1> : k_gemm_proxy_f16x<2, 0, 0>);
}
static const char* name_x(bool streamed, bool odd, bool adaptive = false, bool slots = false, int njb = 8) {
    if (njb < 8 && adaptive && !slots && !odd) {
        if (njb == 1) return streamed ? "fir::k_gemm_proxy_f16x<3, 1, 0, 0, 1>" : "fir::k_gemm_proxy_f16x<3, 0, 0, 0, 1>";
        if (njb == 2) return streamed ? "fir::k_gemm_proxy_f16x<3, 1, 0, 0, 2>" : "fir::k_gemm_proxy_f16x<3, 0, 0, 0, 2>";
    }
    if (adaptive && slots) return streamed ? (odd ? "fir::k_gemm_proxy_f16x<4, 1, 1>" : "fir::k_gemm_proxy_f16x<4, 1, 0>") : (odd ? "fir::k_gemm_proxy_f16x<4, 0, 1>" : "fir::k_gemm_proxy_f16x<4, 0, 0>");
    if (adaptive) return streamed ? (odd ? "fir::k_gemm_proxy_f16x<3, 1, 1>" : "fir::k_gemm_proxy_f16x<3, 1, 0>") : (odd ? "fir::k_gemm_proxy_f16x<3, 0, 1>" : "fir::k_gemm_proxy_f16x<3, 0, 0>");
    return streamed ? (odd ? "fir::k_gemm_proxy_f16x<1, 1, 1>" : "fir::k_gemm_proxy_f16x<1, 1, 0>") : (odd ? "fir::k_gemm_proxy_f16x<1, 0, 1>" : "fir::k_gemm_proxy_f16x<1, 0, 0>");
}

extern "C" {

int fir_gemm_create(fir_gallery* g, fir_gemm** out) { return fir_gemm_create_range(g, FIR_GEMM_F16, 0, out); }

int fir_gemm_create_ex(fir_gallery* g, int32_t precision, fir_gemm** out) { return fir_gemm_create_range(g, precision, 0, out); }

int fir_gemm_create_range(fir_gallery* g, int32_t precision, int32_t end_pos, fir_gemm** out) { return fir_gemm_create_range_ex_(g, precision, end_pos, -1, out); }

void fir_gemm_memory_bytes_(const fir_gemm* m, int64_t* fragments, int64_t* rowmajor, int64_t* scratch) {
    const int64_t rblocks = (std::max<int64_t>(m->v.n, 1) + 31) / 32, np = std::max<int64_t>(m->v.n, 1);
    const int64_t fr = m->gh ? rblocks * m->dk16 * 1024 : m->gb ? rblocks * m->dk16 * 2048 : m->gm ? rblocks * m->dq8 * 1024 : 0;
    const int64_t rm = m->rowmajor ? m->v.n * (int64_t)((m->feat + 3) / 4) * 16 : 0;
    int64_t sc = np * 4 + 16;                                                                     // row norms, their maximum
    sc += 2 * ((int64_t)m->lists_cap * kListCap * 8 + (int64_t)m->lists_cap * 4);                 // candidate lists and counts
    sc += 2 * (m->precision == FIR_GEMM_F32 ? (int64_t)kPasses * (kQT / 32) * m->dq8 * 1024
                                            : (int64_t)kPasses * (kQT / 32) * m->dk16 * (m->precision == FIR_GEMM_BF16_SPLIT ? 2048 : 1024));   // query fragments
    sc += 2 * (int64_t)kPasses * kQT * 4 * 4 + 2 * (int64_t)kRtSubsets * kPasses * kQT * 4;      // per-query scalars, subset minima
    sc += (int64_t)m->ok_cap * 16 + 32;                                                           // certificate flags, the two lists of uncertified queries, their bounds
    if (m->sc_lists) sc += (int64_t)kScQueries * kListCap * 8 + (int64_t)4 * m->dk16 * 1024 + (int64_t)kScQueries * 20;   // one pair's second-chance scratch
    if (m->proxies) sc += (int64_t)m->proxies_nq * np * 4;
    if (fragments) *fragments = fr;
    if (rowmajor) *rowmajor = rm;
    if (scratch) *scratch = sc;
}

int fir_gemm_create_range_ex_(fir_gallery* g, int32_t precision, int32_t end_pos, int32_t rowmajor_mode, fir_gemm** out) {
    if (!g || !out) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    if (precision != FIR_GEMM_F32 && precision != FIR_GEMM_BF16_SPLIT && precision != FIR_GEMM_F16)
        return gemm_fail(FIR_ERR_ARG, "bad precision %d", precision);
    *out = nullptr;
    fir_gemm* m = new (std::nothrow) fir_gemm();
    if (!m) return gemm_fail(FIR_ERR_NOMEM, "host allocation failed");
    m->g = g;
    const void* gp = nullptr;
    if (fir_gallery_view_(g, &m->v) != FIR_OK || fir_gallery_tiled_(g, &gp, &m->dp4) != FIR_OK) { delete m; return gemm_fail(FIR_ERR_ARG, "bad gallery"); }
    m->gal4 = (const float4*)gp;
    m->precision = precision;
    // a prefix [0, end_pos) of every row (the reference's "BF, 64" / "BF, 256" classifiers, ImageTesting.cpp:526-529): its own
    // fp16 fragments and row norms; whole 16-feature k-blocks, fp16 form only
    m->feat = end_pos > 0 && end_pos < m->v.d ? end_pos : m->v.d;
    if (end_pos < 0 || end_pos > m->v.d || (m->feat != m->v.d && (m->feat % 16 != 0 || precision != FIR_GEMM_F16))) {
        const int dd = m->v.d;
        delete m;
        return gemm_fail(FIR_ERR_ARG, "feature prefix [0,%d) of %d: multiples of 16 inside the row, fp16 form only", end_pos, dd);
    }
    if (const char* w = fir_knob_("FIR_GEMM_WIDE")) m->wide = std::atoi(w) != 0;   // experiments: 0 = one pass per gallery read
    m->dq8 = (m->feat + 31) / 32 * 4;     // feature groups of 8, padded to a multiple of 4 groups (zeros)
    m->dk16 = (m->feat + 127) / 128 * 8;  // k-blocks of 16, padded to a multiple of 8 (the paired-pass kernel's double-buffer unit)
    if (precision == FIR_GEMM_F16) m->dk16 = (m->feat + 16 * kRing - 1) / (16 * kRing) * kRing;   // ... to whole double-buffer units
    hipError_t e = hipSetDevice(m->v.device);
    const int64_t rblocks = (std::max<int64_t>(m->v.n, 1) + 31) / 32;
    const int64_t np = std::max<int64_t>(m->v.n, 1);
    if (e == hipSuccess && precision == FIR_GEMM_F32) e = hipMalloc((void**)&m->gm, (size_t)rblocks * m->dq8 * 64 * sizeof(float4));
    if (e == hipSuccess && precision == FIR_GEMM_BF16_SPLIT) e = hipMalloc((void**)&m->gb, (size_t)rblocks * m->dk16 * 128 * sizeof(uint4));
    if (e == hipSuccess && precision == FIR_GEMM_F16) e = hipMalloc((void**)&m->gh, (size_t)rblocks * m->dk16 * 64 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void**)&m->gnorm, (size_t)np * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->gmax, 16);
    for (int b = 0; b < 2; ++b) {
        if (e == hipSuccess && precision == FIR_GEMM_F32) e = hipMalloc((void**)&m->qm[b], (size_t)kPasses * (kQT / 32) * m->dq8 * 64 * sizeof(float4));
        if (e == hipSuccess && precision != FIR_GEMM_F32)
            e = hipMalloc((void**)&m->qbf[b], (size_t)kPasses * (kQT / 32) * m->dk16 * (precision == FIR_GEMM_BF16_SPLIT ? 128 : 64) * sizeof(uint4));
        if (e == hipSuccess) e = hipMalloc((void**)&m->qnorm[b], kPasses * kQT * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&m->qmul[b], kPasses * kQT * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&m->qinv[b], kPasses * kQT * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&m->smin[b], (size_t)kRtSubsets * kPasses * kQT * sizeof(unsigned int));   // top-K: one minimum per subset
        if (e == hipSuccess) e = hipMalloc((void**)&m->tau[b], kPasses * kQT * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&m->awin[b], kPasses * kQT * sizeof(float));
        if (e == hipSuccess) e = hipMalloc((void**)&m->aT[b], (size_t)8 * kPasses * kQT * sizeof(unsigned int));     // (x 8: the K slot words of the K-nearest form)
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->main_done[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->rerank_done[b], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m->prep_done[b], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&m->queries_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&m->copy, hipStreamNonBlocking);
    for (int b = 0; b < 2 && e == hipSuccess; ++b) e = hipEventCreateWithFlags(&m->copy_done[b], hipEventDisableTiming);
    m->sample_rows = (int)std::min<int64_t>(np, std::max<int64_t>(kMinSampleRows, np / 64));
    m->rt_sample_rows = (int)std::min<int64_t>(np, std::max<int64_t>(kMinSampleRows, np / 32));    // n/16 .. n/48 measured: 1.022 / 1.038 / 1.028 M queries/s at 1M x 512
    if (const char* w = fir_knob_("FIR_GEMM_SAMPLE_DIV"))      // experiments
        m->rt_sample_rows = (int)std::min<int64_t>(np, std::max<int64_t>(kMinSampleRows, np / std::max(1, std::atoi(w))));
    if (e == hipSuccess) e = hipMalloc((void**)&m->fb_state, kFbStateWords * sizeof(int));
    if (e == hipSuccess) e = hipMemset(m->fb_state, 0, kFbStateWords * sizeof(int));
    if (e == hipSuccess) e = hipMalloc((void**)&m->sc_qnorm, kScQueries * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->sc_qmul, kScQueries * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->sc_qinv, kScQueries * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->sc_tau, kScQueries * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->sc_counts, kScQueries * sizeof(int));
    if (e == hipSuccess && precision == FIR_GEMM_F16) e = hipMalloc((void**)&m->sc_lists, (size_t)kScQueries * kListCap * sizeof(unsigned long long));
    if (e == hipSuccess && precision == FIR_GEMM_F16) e = hipMalloc((void**)&m->sc_qbf, (size_t)4 * m->dk16 * 64 * sizeof(uint4));
    {   // the exact device scan of the uncertified rest keeps eight queries in LDS
        m->fb_lds = (size_t)m->dp4 * 4 * 8 * sizeof(float);
        if (m->fb_lds > kRerankLdsMax) {
            const int dd = m->v.d;
            fir_gemm_destroy(m);
            return gemm_fail(FIR_ERR_ARG, "rows of %d features are too long for the matrix-core path's exact fallback", dd);
        }
        if (e == hipSuccess && m->fb_lds > 48 * 1024)
            e = hipFuncSetAttribute((const void*)k_gemm_exact_fb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRerankLdsMax);
        const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / std::max<size_t>(m->fb_lds, 1)));
        m->fb_grid = std::max(1, m->v.cus) * per_cu;
    }
    const int lds_bytes = precision == FIR_GEMM_F32 ? (kQT / 32) * std::min(m->dq8, kSlab8) * 64 * (int)sizeof(float4)
                                                    : (kQT / 32) * std::min(m->dk16, kSlab16) * 128 * (int)sizeof(uint4);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_bf16<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_bf16<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_bf16_wide, hipFuncAttributeMaxDynamicSharedMemorySize, kWideLds);
    {   // the re-rank keeps the query and up to kRerankGroup candidate rows in LDS
        const size_t row_bytes = (size_t)(m->dp4 + 1) * sizeof(float4);
        size_t want_group = kRerankGroup;
        if (const char* w = fir_knob_("FIR_GEMM_RERANK_GROUP")) want_group = (size_t)std::max(1, std::min(64, std::atoi(w)));      // experiments
        m->rerank_group = (int)std::min<size_t>(want_group, kRerankLdsMax / row_bytes > 1 ? kRerankLdsMax / row_bytes - 1 : 0);
        if (m->rerank_group < 1) { delete m; return gemm_fail(FIR_ERR_ARG, "rows of %d features are too long for the matrix-core path's re-rank", m->v.d); }
        if (e == hipSuccess && (size_t)(m->rerank_group + 1) * row_bytes > 48 * 1024)
            e = hipFuncSetAttribute((const void*)k_gemm_rerank, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRerankLdsMax);
        if (e == hipSuccess && (size_t)(m->rerank_group + 1) * row_bytes > 48 * 1024)
            e = hipFuncSetAttribute((const void*)k_gemm_rerank_topk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRerankLdsMax);
    }
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
#define FIR_RT_ATTR(D) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16_regtile<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RegTile<D>::lds_bytes); \
                       if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16_regtile<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RegTile<D>::lds_bytes);
    FIR_RT_ATTR(8) FIR_RT_ATTR(16) FIR_RT_ATTR(32)
#undef FIR_RT_ATTR
#define FIR_X_ATTR(M, S, O) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16x<M, S, O>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
    FIR_X_ATTR(1, 0, 0) FIR_X_ATTR(1, 0, 1) FIR_X_ATTR(1, 1, 0) FIR_X_ATTR(1, 1, 1) FIR_X_ATTR(2, 0, 0) FIR_X_ATTR(2, 0, 1) FIR_X_ATTR(2, 1, 0) FIR_X_ATTR(2, 1, 1)
    FIR_X_ATTR(3, 0, 0) FIR_X_ATTR(3, 0, 1) FIR_X_ATTR(3, 1, 0) FIR_X_ATTR(3, 1, 1)
    FIR_X_ATTR(4, 0, 0) FIR_X_ATTR(4, 0, 1) FIR_X_ATTR(4, 1, 0) FIR_X_ATTR(4, 1, 1)
#undef FIR_X_ATTR
    m->mfma16 = precision == FIR_GEMM_F16;
    if (const char* w = fir_knob_("FIR_GEMM_MFMA16")) m->mfma16 = std::atoi(w) != 0 && precision == FIR_GEMM_F16;
    if (const char* w = fir_knob_("FIR_GEMM_STAGGER")) m->stagger = std::atoi(w);     // 1 = half a unit, 2 = half a row block at 512 features
    if (const char* w = fir_knob_("FIR_GEMM_PRIO")) m->prio = std::atoi(w) != 0;
    if (const char* w = fir_knob_("FIR_GEMM_NO_BLOCK_BOUND")) m->no_block_bound = std::atoi(w) != 0;
#ifdef FIR_AUDIT      // knobs that change answers: the audit build only (libfir_amd_audit.so; fir_internal.h)
    if (const char* w = fir_knob_("FIR_GEMM_DBG_SKIP")) m->dbg_skip = std::atoi(w) & 1023;
    if (const char* w = fir_knob_("FIR_GEMM_DBG_BLOCK")) m->dbg_block = std::atoi(w) == 256 ? 256 : kGemmBlock;     // timing experiments only: the answers are wrong
    if (const char* w = fir_knob_("FIR_GEMM_EREL_SCALE")) m->erel_scale = (float)std::atof(w);
#endif
    if (const char* w = fir_knob_("FIR_GEMM_ADAPTIVE")) m->adaptive = std::atoi(w);
    if (const char* w = fir_knob_("FIR_GEMM_ADAPTIVE_TOPK")) m->adaptive_topk = std::atoi(w) != 0;
    if (const char* w = fir_knob_("FIR_GEMM_SHARE_STREAMED")) m->share_streamed = std::max(1, std::min(16, std::atoi(w)));
    if (const char* w = fir_knob_("FIR_GEMM_FEW_BLOCKS")) m->few_blocks = std::atoi(w) != 0;
    if (const char* w = fir_knob_("FIR_GEMM_SHARE")) m->share_max = std::max(0, std::min(16, std::atoi(w)));
    // the 16-row kernels always run the smallest-proxy sample flow with its XCD-shared launches: one workgroup per CU, CUs in eights
    if (m->mfma16 && !(m->share_max > 0 && (m->v.cus & 7) == 0 && m->v.cus >= 8)) m->mfma16 = 0;
    if (const char* w = fir_knob_("FIR_GEMM_REGTILE")) m->regtile = std::atoi(w);
    if (const char* w = fir_knob_("FIR_GEMM_STREAMED")) m->streamed = std::atoi(w);   // experiments: 0 / 1 force the form, -1 = by row length
    if (e == hipSuccess && m->v.n > 0) {
        // row norms come from the f32 packer (run on a one-group scratch when only the bf16 fragments are kept)
        if (precision == FIR_GEMM_BF16_SPLIT) {
            const int64_t totalb = rblocks * m->dk16 * 64;
            hipLaunchKernelGGL(k_gemm_pack_gallery_bf16, dim3((unsigned)((totalb + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4,
                               m->dk16, m->gb);
            hipLaunchKernelGGL(k_gemm_row_norms, dim3((unsigned)((m->v.n + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4, m->gnorm, (m->feat + 3) / 4);
        }
        if (precision == FIR_GEMM_F16) {
            // one power-of-two scale for the whole gallery: its largest |value| lands in [2^13, 2^14)
            float h_max = 0.f;
            e = hipMemsetAsync(m->gmax, 0, 16, m->v.stream);
            const int64_t count4 = (int64_t)((m->v.n + 63) / 64) * m->dp4 * 64;
            hipLaunchKernelGGL(k_gemm_absmax, dim3((unsigned)std::min<int64_t>((count4 + 255) / 256, 4096)), dim3(256), 0, m->v.stream, m->gal4, count4, m->gmax);
            if (e == hipSuccess) e = hipMemcpyAsync(&h_max, m->gmax, sizeof(float), hipMemcpyDeviceToHost, m->v.stream);
            if (e == hipSuccess) e = hipStreamSynchronize(m->v.stream);
            int ex = 0;
            if (h_max > 0.f && h_max < __builtin_huge_valf()) { (void)std::frexp(h_max, &ex); m->gallery_exp = 14 - ex; }
            else m->gallery_exp = h_max > 0.f ? 1000 : 0;      // non-finite gallery value: every query is left to the exact scan
            const float scale = (m->gallery_exp >= -100 && m->gallery_exp <= 100) ? std::ldexp(1.0f, m->gallery_exp) : 0.f;
            const int64_t totalh = rblocks * m->dk16 * 64;
            if (m->mfma16)
                hipLaunchKernelGGL(k_gemm_pack_gallery_f16x, dim3((unsigned)((totalh + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4,
                                   m->dk16, scale, m->gh, m->feat == m->v.d ? m->dp4 * 4 : m->feat);
            else
            hipLaunchKernelGGL(k_gemm_pack_gallery_f16, dim3((unsigned)((totalh + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4,
                               m->dk16, scale, m->gh, m->feat == m->v.d ? m->dp4 * 4 : m->feat);
            hipLaunchKernelGGL(k_gemm_row_norms, dim3((unsigned)((m->v.n + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4, m->gnorm, (m->feat + 3) / 4);
        }
        const int64_t total = precision == FIR_GEMM_F32 ? rblocks * m->dq8 * 64 : 0;
        if (total > 0)
        hipLaunchKernelGGL(k_gemm_pack_gallery, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, m->v.stream, m->gal4, m->v.n, m->dp4, m->dq8,
                           m->gm, m->gnorm);
        hipLaunchKernelGGL(k_gemm_max, dim3(1), dim3(256), 0, m->v.stream, m->gnorm, m->v.n, m->gmax);
        // The re-rank's row-major copy of the compared features, when it leaves at least three times its size of HBM free
        // (FIR_GEMM_ROWMAJOR=0 / 1: never / whenever it can be allocated)
        {
            const int d4 = (m->feat + 3) / 4;
            const size_t want = (size_t)m->v.n * d4 * sizeof(float4);
            size_t free_b = 0, total_b = 0;
            int mode = rowmajor_mode;
            if (const char* w = fir_knob_("FIR_GEMM_ROWMAJOR")) { if (mode < 0) mode = std::atoi(w); }
            const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b >= 4 * want;
            if (mode != 0 && (room || mode > 0) && hipMalloc((void**)&m->rowmajor, want) == hipSuccess) {
                const int64_t tiles = (m->v.n + 63) / 64;
                for (int64_t t0 = 0; t0 < tiles; t0 += 65535 * 16) {      // gridDim.x limit: slabs of tiles
                    const int64_t nt = std::min<int64_t>(tiles - t0, 65535 * 16);
                    hipLaunchKernelGGL(k_gemm_untile, dim3((unsigned)nt, (unsigned)((d4 + 31) / 32)), dim3(256), 0, m->v.stream, m->gal4 + (size_t)t0 * m->dp4 * 64,
                                       m->v.n - t0 * 64, m->dp4, d4, m->rowmajor + (size_t)t0 * 64 * d4);
                }
            } else {
                (void)hipGetLastError();
                m->rowmajor = nullptr;
            }
        }
        e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(m->v.stream);
    }
    if (e != hipSuccess) {
        const int rc = gemm_fail(e == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP, "GEMM-path setup: %s", hipGetErrorString(e));
        fir_gemm_destroy(m);
        return rc;
    }
    *out = m;
    return FIR_OK;
}

int fir_gemm_destroy(fir_gemm* m) {
    if (!m) return FIR_OK;
    (void)hipSetDevice(m->v.device);
    if (m->v.stream || !m->f64) (void)hipStreamSynchronize(m->v.stream);
    if (m->side) { (void)hipStreamSynchronize(m->side); (void)hipStreamDestroy(m->side); }
    for (int b = 0; b < 2; ++b) {
        (void)hipFree(m->qm[b]); (void)hipFree(m->qbf[b]); (void)hipFree(m->qnorm[b]); (void)hipFree(m->qmul[b]); (void)hipFree(m->qinv[b]); (void)hipFree(m->smin[b]); (void)hipFree(m->tau[b]); (void)hipFree(m->awin[b]); (void)hipFree(m->aT[b]); (void)hipFree(m->lists[b]); (void)hipFree(m->counts[b]);
        if (m->main_done[b]) (void)hipEventDestroy(m->main_done[b]);
        if (m->rerank_done[b]) (void)hipEventDestroy(m->rerank_done[b]);
        if (m->prep_done[b]) (void)hipEventDestroy(m->prep_done[b]);
    }
    if (m->queries_ready) (void)hipEventDestroy(m->queries_ready);
    (void)hipFree(m->gm); (void)hipFree(m->gb); (void)hipFree(m->gh); (void)hipFree(m->rowmajor); (void)hipFree(m->proxies); (void)hipFree(m->gnorm); (void)hipFree(m->gmax); (void)hipFree(m->sample); (void)hipFree(m->ok);
    (void)hipFree(m->fb_state); (void)hipFree(m->fb_list); (void)hipFree(m->fb_list2); (void)hipFree(m->fb_tau2);
    (void)hipFree(m->sc_qnorm); (void)hipFree(m->sc_qmul); (void)hipFree(m->sc_qinv); (void)hipFree(m->sc_tau); (void)hipFree(m->sc_counts);
    (void)hipFree(m->sc_lists); (void)hipFree(m->sc_qbf);
        if (m->copy) { (void)hipStreamSynchronize(m->copy); (void)hipStreamDestroy(m->copy); }
    for (int b = 0; b < 2; ++b) if (m->copy_done[b]) (void)hipEventDestroy(m->copy_done[b]);
    delete m;
    return FIR_OK;
}

// out[0] = 64-query passes queued so far, out[1] = queries whose first certificate did not hold (they took a second matrix-core
// pass or, beyond its rounds, went straight to the exact device scan), out[2] = queries the exact device scan answered. The two
// counters live on the device (fir_gemm_fb.h): this call waits for the device's work and reads them.
int fir_gemm_stats_ex(const fir_gemm* m, int64_t out[3]) {
    if (!m || !out) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    GEMM_HIP(hipSetDevice(m->v.device));
    GEMM_HIP(hipDeviceSynchronize());
    unsigned long long tot[2] = {0, 0};
    GEMM_HIP(hipMemcpy(tot, m->fb_state + 4, sizeof tot, hipMemcpyDeviceToHost));
    out[0] = m->passes;
    out[1] = (int64_t)tot[0];
    out[2] = (int64_t)tot[1];
    return FIR_OK;
}

// What the first (up to 8) queries whose FIRST certificate did not hold looked like, since this state was created: per query four
// floats -- entries its candidate list was asked to take (4096 fit), the bound the pass appended below (|g|^2 - 2 q.g units), the
// smallest stored proxy (the K-th smallest for the K nearest), |q|^2. *count = how many of the 8 slots are filled. Diagnostics for
// a benchmark line: a bound of +inf with a million entries is a pass that never found its threshold, a finite one with a few
// thousand a threshold that stayed loose.
int fir_gemm_uncertified_notes(const fir_gemm* m, float out[32], int32_t* count) {
    if (!m || !out || !count) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    GEMM_HIP(hipSetDevice(m->v.device));
    GEMM_HIP(hipDeviceSynchronize());
    int h[kFbStateWords];
    GEMM_HIP(hipMemcpy(h, m->fb_state, sizeof h, hipMemcpyDeviceToHost));
    *count = h[2] < 8 ? h[2] : 8;
    std::memcpy(out, h + 8, 32 * sizeof(float));
    return FIR_OK;
}

int fir_gemm_stats(const fir_gemm* m, int64_t* passes, int64_t* fallback_queries) {
    int64_t o[3];
    const int rc = fir_gemm_stats_ex(m, o);
    if (rc) return rc;
    if (passes) *passes = o[0];
    if (fallback_queries) *fallback_queries = o[2];
    return FIR_OK;
}

}  // extern "C"

// Everything a call still owes its uncertified queries, queued on `st` behind the last re-rank (fir_gemm_fb.h): second-chance rounds
// (sc_rounds > 0: the 16-row fp16 flow only), the collection of what is left, the exact device scan (K launches for the K nearest).
static int gemm_finish_(fir_gemm* m, const float* d_queries, int k, uint64_t* d_keys, hipStream_t st, float e_rel, int sc_rounds) {
    const int d = m->feat, qs = m->v.d;
    const int64_t n = m->v.n;
    const bool streamed = m->dk16 > kSlabH || m->streamed > 0;
    const size_t rr_lds = (size_t)(m->rerank_group + 1) * (m->dp4 + 1) * sizeof(float4);
    for (int r = 0; r < sc_rounds; ++r) {
        const int off = r * kScQueries;
        hipLaunchKernelGGL(k_gemm_sc_prep, dim3(kScQueries), dim3(64), 0, st, d_queries, d, qs, m->gallery_exp, (const int*)m->fb_state, (const int*)m->fb_list, off,
                           (const float*)m->fb_tau2, m->sc_qnorm, m->sc_qmul, m->sc_qinv, m->sc_tau, m->sc_counts);
        hipLaunchKernelGGL(k_gemm_pack_queries_f16x, dim3((4 * m->dk16 * 64 + 255) / 256, 1), dim3(256), 0, st, d_queries, kScQueries, d, m->dk16, (const float*)m->sc_qmul,
                           m->sc_qbf, qs, (const int*)m->fb_list + off, (const int*)m->fb_state, off);
        // one pair over all CUs (share = 1: 256 row ranges), the gallery stream read once (nt)
        hipLaunchKernelGGL(pick_x(1, streamed, (m->dk16 / kRing) & 1), dim3(m->v.cus, 1), dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->sc_qbf, (const float*)m->sc_qinv, n,
                           (int64_t)0, n, m->dk16, (const float*)m->sc_tau, m->sc_lists, m->sc_counts, (float*)nullptr, 0, 1, 1 | (m->no_block_bound ? 64 : 0), 1,
                           (unsigned int*)m->fb_state, off);
        const RerankFb fb = {m->fb_state, nullptr, nullptr, 0, (const int*)m->fb_list + off, off};
        if (k == 1)
            hipLaunchKernelGGL(k_gemm_rerank, dim3(kScQueries), dim3(64), rr_lds, st, m->sc_lists, m->sc_counts, m->sc_tau, m->gal4, d_queries, m->sc_qnorm, m->gmax, n, d,
                               m->dp4, m->v.row_offset, e_rel, m->rerank_group, (unsigned long long*)d_keys, m->ok, qs, m->rowmajor, fb);
        else
            hipLaunchKernelGGL(k_gemm_rerank_topk, dim3(kScQueries), dim3(64), rr_lds, st, m->sc_lists, m->sc_counts, m->sc_tau, m->gal4, d_queries, m->sc_qnorm, m->gmax, n,
                               d, m->dp4, m->v.row_offset, e_rel, m->rerank_group, k, (unsigned long long*)d_keys, m->ok, qs, m->rowmajor, fb);
    }
    hipLaunchKernelGGL(k_gemm_fb_collect, dim3(1), dim3(256), 0, st, m->fb_state, (const int*)m->fb_list, (const int*)m->ok, m->fb_list2, (unsigned long long*)d_keys, k);
    for (int r = 0; r < k; ++r)
        hipLaunchKernelGGL(k_gemm_exact_fb, dim3(m->fb_grid), dim3(256), m->fb_lds, st, m->gal4, n, m->dp4, d, m->v.row_offset, d_queries, qs, (const int*)m->fb_state,
                           (const int*)m->fb_list2, (unsigned long long*)d_keys, k, r);
    GEMM_HIP(hipGetLastError());
    return FIR_OK;
}

// the per-call buffers of the uncertified-query lists (and the certificate flags), grown to the largest call seen
static int gemm_fb_reserve_(fir_gemm* m, int32_t qb) {
    if ((size_t)qb <= m->ok_cap) return FIR_OK;
    GEMM_HIP(hipDeviceSynchronize());                        // (an earlier call's kernels may still read the old ones)
    (void)hipFree(m->ok); (void)hipFree(m->fb_list); (void)hipFree(m->fb_list2); (void)hipFree(m->fb_tau2);
    m->ok = nullptr; m->fb_list = nullptr; m->fb_list2 = nullptr; m->fb_tau2 = nullptr;
    m->ok_cap = 0;
    const size_t cap = (size_t)std::max(qb, 1024);
    GEMM_HIP(hipMalloc((void**)&m->ok, cap * sizeof(int)));
    GEMM_HIP(hipMalloc((void**)&m->fb_list, cap * sizeof(int)));
    GEMM_HIP(hipMalloc((void**)&m->fb_list2, cap * sizeof(int)));
    GEMM_HIP(hipMalloc((void**)&m->fb_tau2, cap * sizeof(float)));
    m->ok_cap = cap;
    return FIR_OK;
}

// d_queries / d_keys: device pointers. Everything is queued on `stream` (and the handle's side streams, joined before the call
// returns control of `stream`): no host synchronisation -- the keys are final in stream order.

// The K nearest rows of every query, K = 1 (fir_gemm_search_top1_keys_dev) or 2..kTopKMax (fir_gemm_search_topk_keys_dev):
// d_keys[q * k + r], ascending.
static int gemm_search(fir_gemm* m, const float* d_queries, int32_t qb, int k, uint64_t* d_keys, void* stream, const float* h_queries = nullptr) {
    // h_queries: the queries are still in host memory; d_queries is the device buffer they go to, one super-batch at a time,
    // each upload queued (on its own stream) right before that super-batch's preparation
    if (!m || !d_keys || (qb > 0 && !d_queries)) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return gemm_fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > kTopKMax) return gemm_fail(FIR_ERR_ARG, "k=%d outside [1,%d]", k, kTopKMax);
    if (qb == 0) return FIR_OK;
    GEMM_HIP(hipSetDevice(m->v.device));
    hipStream_t st = stream ? (hipStream_t)stream : m->v.stream;
    const int d = m->feat;          // features compared
    const int qs = m->v.d;          // floats between consecutive queries (and gallery rows): the whole row
    const int64_t n = m->v.n;
    if (n == 0 && h_queries) GEMM_HIP(hipMemcpyAsync((void*)d_queries, h_queries, (size_t)qb * qs * sizeof(float), hipMemcpyHostToDevice, st));
    if (n == 0)
        return k == 1 ? fir_search_top1_exact_keys_dev_(m->g, d_queries, qb, 0, d, d_keys, st)
                      : fir_search_topk_exact_keys_dev_(m->g, d_queries, qb, d, k, d_keys, st);
    {
        const int rcr = gemm_fb_reserve_(m, qb);
        if (rcr) return rcr;
    }
    // this call's two list lengths (the fp16 forms clear them in the first super-batch's query preparation: one launch less per call)
    if (m->precision != FIR_GEMM_F16) GEMM_HIP(hipMemsetAsync(m->fb_state, 0, 2 * sizeof(int), st));
    const size_t lds = m->precision == FIR_GEMM_F32 ? (size_t)(kQT / 32) * std::min(m->dq8, kSlab8) * 64 * sizeof(float4)
                                                    : (size_t)(kQT / 32) * std::min(m->dk16, kSlab16) * 128 * sizeof(uint4);
    // fp16: both operands rounded to 11 bits -> |q~.g~ - q.g| <= (2^-10 + 2^-22) sum|q_k g_k| + the sub-normal tails
    // (< 2^-27 |q||g| for d <= 2^20), doubled in p and with |q||g| <= (|q|^2 + max|g|^2) / 2: 2^-10 (1 + 2^-4) covers it
    const float e_rel = m->erel_scale * (8.0f * (float)d * 5.9604645e-8f +
                        (m->precision == FIR_GEMM_BF16_SPLIT ? 6.1035156e-5f : m->precision == FIR_GEMM_F16 ? 9.765625e-4f * 1.0625f : 0.0f));
    const int grid = m->v.cus;      // one 512-thread workgroup per CU
    const int sample_rows = m->sample_rows;
    const int sample_grid = (sample_rows + 63) / 64;
    // super-batch = the queries of ONE set of launches (preparation, full passes, re-rank): up to kPasses passes of 64 queries. Large
    // sets amortise the boundaries between full passes (where the small kernels run: they cannot share a CU with a full-pass
    // workgroup), but a call needs a few super-batches for its preparation / re-rank to overlap anything: a quarter of the call,
    // in whole 1 024-query launches
    // Galleries whose 1 024-query launch is short (n d <= 1.5e8: under ~0.25 ms) take the whole call (up to kPasses passes) as ONE
    // super-batch: between two super-batches the next one's preparation queues behind the previous one's re-rank, which runs squeezed
    // between the full pass's workgroups (100 000 x 512, 4 096 queries: ~65 us of bubble per boundary and a re-rank four times
    // slower than alone, for 110 us launches -- profiles/r03_small_call_timeline.txt)
    const bool short_launches = (double)n * (double)d <= 1.5e8 && !fir_knob_("FIR_GEMM_QUARTERS");
    const int sbq = short_launches ? std::min(kPasses * kQT, std::max(1024, (qb + 1023) / 1024 * 1024))
                                   : std::min(kPasses * kQT, std::max(1024, (qb / 4 + 1023) / 1024 * 1024));
    const int nsb = (qb + sbq - 1) / sbq;
    {   // candidate lists and counts: sized by the super-batch in use (whole 128-query pairs), not by the largest one possible --
        // a cache-resident gallery that gets 128-query calls holds 8 MiB of lists, not 512
        const int need = (std::min(sbq, qb) + 2 * kQT - 1) / (2 * kQT) * (2 * kQT);
        if (need > m->lists_cap) {
            GEMM_HIP(hipStreamSynchronize(st));
            GEMM_HIP(hipStreamSynchronize(m->side));
            for (int b = 0; b < 2; ++b) {
                (void)hipFree(m->lists[b]); (void)hipFree(m->counts[b]);
                m->lists[b] = nullptr; m->counts[b] = nullptr;
            }
            m->lists_cap = 0;
            for (int b = 0; b < 2; ++b) {
                GEMM_HIP(hipMalloc((void**)&m->lists[b], (size_t)need * kListCap * sizeof(unsigned long long)));
                GEMM_HIP(hipMalloc((void**)&m->counts[b], (size_t)need * sizeof(int)));
            }
            m->lists_cap = need;
        }
    }
    // Two streams. `st` carries only the full passes over the gallery, back to back; `side` carries everything small:
    // the preparation of super-batch i+1 (query norms / scales / fragments, the sample pass, tau) and the exact re-rank +
    // certificate of super-batch i, both under super-batch i's (or i+1's) full pass. Order on `side`:
    // prep(0) prep(1) rerank(0) prep(2) rerank(1) ... -- prep(i+2) reuses the scratch set rerank(i) has just finished with.
    // (a call of ONE super-batch whose preparation runs on `st` has nothing to put under anything: its re-rank follows its pass on `st`
    // as well -- every hop to the side stream and back is 10-12 us of a 300-us call)
    const char* sp_env = fir_knob_("FIR_GEMM_SERIAL_PREP");
    const bool serial_prep = !(sp_env && std::atoi(sp_env) == 0) && !h_queries;        // (see the note on prep() below)
    const bool one_stream = nsb == 1 && serial_prep && !fir_knob_("FIR_GEMM_TWO_STREAMS");
    if (!one_stream) {
        GEMM_HIP(hipEventRecord(m->queries_ready, st));
        GEMM_HIP(hipStreamWaitEvent(m->side, m->queries_ready, 0));
    }
    // the register-tile flow (fir_gemm_regtile.h) for fp16 galleries whose rows fit the compute waves' registers
    typedef void (*rt_fn)(const uint4*, const float*, const uint4*, const float*, int64_t, int64_t, int, const float*, unsigned long long*, int*, unsigned int*, int, int, int);
    rt_fn rt_main = nullptr, rt_sample = nullptr;
    size_t rt_lds = 0;
    if (m->precision == FIR_GEMM_F16 && !m->mfma16 && m->share_max > 0 && (grid & 7) == 0 && grid >= 8) {
#define FIR_RT_PICK(D) if (m->dk16 == D) { rt_main = k_gemm_proxy_f16_regtile<D, false>; rt_sample = k_gemm_proxy_f16_regtile<D, true>; rt_lds = RegTile<D>::lds_bytes; }
        FIR_RT_PICK(8) FIR_RT_PICK(16) FIR_RT_PICK(32)
#undef FIR_RT_PICK
    }
    // sample pass + tau through the register-tile kernel: the smallest sampled proxy + one window (top-1), the K-th smallest of 64
    // disjoint subsets' minima + one window (top-K); other row lengths: the block-minimum sample + k_gemm_tau
    // the same flow on the 16-row fragment order: k_gemm_proxy_f16x<2, *> is its sample pass, for any row length
    const bool x_flow = m->precision == FIR_GEMM_F16 && m->mfma16 && m->share_max > 0 && (grid & 7) == 0 && grid >= 8;
    const bool rt_flow = rt_main != nullptr || x_flow;
    // Which super-batches find their threshold on the way. Until round 4's second half the answer depended on how many row groups a
    // workgroup sees ((row groups) * (pairs per launch) / CUs: below a dozen the bound is still loose when the workgroup ends, 1 000-2 400
    // appended rows per query at 100k x 512 with 256 queries) and such super-batches kept the sample flow. Since an appended row no longer
    // waits for the gallery prefetch and a wave's first row block is summed once, the pass on the way is never slower for top-1 and
    // 10-23 % faster wherever the sample flow used to run (8 192 .. 700 000 rows x 128 .. 4 096 queries, no second pass anywhere:
    // profiles/r04_adaptive_cutoff.txt): top-1 always takes it. The K-nearest form (eight slots per query fill more slowly) wins from eight
    // pairs per launch on and on cache-sized galleries, and loses 3-35 % with one or two pairs over 100 000+ rows: those keep the sample flow.
    auto adaptive_for = [&](int nq_sb) -> bool {
        if (!(x_flow && m->adaptive > 0 && (k == 1 || m->adaptive_topk))) return false;
        if (m->adaptive > 1) return true;                                   // FIR_GEMM_ADAPTIVE=2: always
        const int pairs_sb = ((nq_sb + kQT - 1) / kQT + 1) / 2;
        const bool str = m->dk16 > kSlabH || m->streamed > 0;
        int P = 1;
        while (P * 2 <= pairs_sb && P * 2 <= (str ? std::min(m->share_max, m->share_streamed) : m->share_max)) P *= 2;
        const int64_t row_groups = ((n + 31) / 32 + kGemmBlock / 64 - 1) / (kGemmBlock / 64);
        // (the K-nearest form's eight slots per query fill more slowly: 256 queries against 1M x 512 measured 479 k q/s against the sample
        // flow's 514 k, 4 096 queries 1.02 M against 0.92 M -- profiles/r04_topk_slots.txt)
        if (k == 1) return true;
        return P >= 8 || n <= 65536 || row_groups * P / std::max(grid, 1) >= 100;
    };
#ifdef FIR_AUDIT
    // 1 = no refresh, 2 = no exchange of the adaptive bound between workgroups (still sound: a looser bound appends more -- this is how
    // the tests drive lists into overflow and the second-chance pass on benign data)
    const int adapt_dbg = fir_knob_("FIR_GEMM_ADAPT_DBG") ? (std::atoi(fir_knob_("FIR_GEMM_ADAPT_DBG")) & 3) << 2 : 0;
#else
    const int adapt_dbg = 0;
#endif
    const int sub_stride = k > 1 ? kPasses * kQT : 0;      // top-K: the sample as kRtSubsets subset minima per query (k_gemm_tau_kmin)
    // the full pass: both kernels run at ~1 KiB of LDS traffic per MFMA and within 7 % of each other (profiles/r02_gemm_kernel_choice.txt):
    // register tile ahead up to 256 features, LDS tile ahead at 512
    const bool rt_full = rt_main != nullptr && (m->regtile > 0 || (m->regtile < 0 && m->dk16 <= 16));
    if (!rt_flow && !m->sample) {
        // the sample of the order-statistic flow, on first use: one block minimum per 32 sampled rows and query (fp16), every proxy (f32 / bf16)
        const size_t per_query = m->precision == FIR_GEMM_F16 ? (size_t)(m->sample_rows + 31) / 32 : (size_t)m->sample_rows;
        GEMM_HIP(hipMalloc((void**)&m->sample, (size_t)kPasses * kQT * per_query * sizeof(float)));
    }
    // A super-batch's preparation (query packing, the top-K sample pass) runs on the MAIN stream right in front of its full passes: alone on
    // the chip it is 25 us (top-K: + 0.3 ms of sample passes); on the side stream, "under" the previous super-batch's full passes, its small
    // kernels only got the CUs a full-pass workgroup had just left and slowed those passes down (1M x 512, 32 768 queries per call: top-1
    // 1.253 -> 1.296 M q/s, top-5 1.018 -> 1.071 M; FIR_GEMM_SERIAL_PREP=0 is the old placement). Host-pointer calls keep the side stream:
    // there the preparation waits for the super-batch's upload, which is what overlaps the passes.
    auto prep = [&](int sb) -> int {
        hipStream_t ps = serial_prep ? st : m->side;
        if (serial_prep && sb >= 2) GEMM_HIP(hipStreamWaitEvent(st, m->rerank_done[sb & 1], 0));      // the re-rank of sb - 2 read this buffer
        const int q0 = sb * sbq;
        const int nq = std::min(sbq, qb - q0);
        if (h_queries) {
            GEMM_HIP(hipMemcpyAsync((void*)(d_queries + (size_t)q0 * qs), h_queries + (size_t)q0 * qs, (size_t)nq * qs * sizeof(float), hipMemcpyHostToDevice, m->copy));
            GEMM_HIP(hipEventRecord(m->copy_done[sb & 1], m->copy));
            GEMM_HIP(hipStreamWaitEvent(ps, m->copy_done[sb & 1], 0));
        }
        const int np = (nq + kQT - 1) / kQT;
        const int b = sb & 1;
        const float* dq = d_queries + (size_t)q0 * qs;
        if (m->precision == FIR_GEMM_F16) {
            const int pairs = (np + 1) / 2;                      // 128 queries per gallery read; a half-filled pair is zero-padded
            // (the candidate counts are cleared and, for the adaptive pass, its per-query state is set by the same launch)
            const bool adaptive_prep = adaptive_for(nq);
            hipLaunchKernelGGL(k_gemm_qprep_f16, dim3(pairs * 2 * kQT), dim3(64), 0, ps, dq, nq, d, m->gallery_exp, m->qnorm[b], m->qmul[b], m->qinv[b], qs,
                               m->counts[b], adaptive_prep ? m->awin[b] : (float*)nullptr, adaptive_prep ? m->aT[b] : (unsigned int*)nullptr,
                               (const float*)m->gmax, e_rel, k > 1 ? k : 0, (unsigned int*)nullptr, sb == 0 ? m->fb_state : (int*)nullptr);
            if (m->mfma16)
                hipLaunchKernelGGL(k_gemm_pack_queries_f16x, dim3((4 * m->dk16 * 64 + 255) / 256, pairs), dim3(256), 0, ps, dq, nq, d, m->dk16, m->qmul[b],
                                   m->qbf[b], qs);
            else
            hipLaunchKernelGGL(k_gemm_pack_queries_f16, dim3((4 * m->dk16 * 64 + 255) / 256, pairs), dim3(256), 0, ps, dq, nq, d, m->dk16, m->qmul[b],
                               m->qbf[b], qs);
            const bool adaptive = adaptive_for(nq);
            if (adaptive) {
                // no sample pass: the full pass finds its threshold on the way (k_gemm_proxy_f16x<3, *>)
                // (k_gemm_qprep_f16 has set the windows and start values)
            } else if (rt_flow) {
                // the smallest proxy of a row sample per query (register-tile kernel over rows [0, rt_sample_rows)), tau = that + one window
                GEMM_HIP(hipMemsetD32Async((hipDeviceptr_t)m->smin[b], (int)0xFF800000u, sub_stride ? (size_t)kRtSubsets * sub_stride : (size_t)pairs * 2 * kQT, ps));
                for (int p0 = 0; p0 < pairs;) {
                    int P = 1;
                    while (P * 2 <= pairs - p0 && P * 2 <= m->share_max) P *= 2;
                    const size_t qo = (size_t)p0;
                    // every rb_stride-th row block: the sample is spread over the whole gallery
                    const int64_t sample_blocks = ((int64_t)m->rt_sample_rows + 31) / 32;
                    const int rb_stride = (int)std::max<int64_t>(1, ((n + 31) / 32) / sample_blocks);
                    if (x_flow) {
                        const bool xs = m->dk16 > kSlabH || m->streamed > 0;
                        hipLaunchKernelGGL(pick_x(2, xs, (m->dk16 / kRing) & 1), dim3(grid), dim3(kGemmBlock), kHalfLds, ps, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64,
                                           m->qinv[b] + qo * 2 * kQT, n, (int64_t)0, sample_blocks * 32, m->dk16, m->tau[b], m->lists[b], m->counts[b], (float*)nullptr,
                                           0, P, P <= 1 ? 1 : 0, rb_stride, m->smin[b] + qo * 2 * kQT, sub_stride);
                    } else
                    hipLaunchKernelGGL(rt_sample, dim3(grid), dim3(512), rt_lds, ps, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT, n,
                                       sample_blocks * 32, rb_stride, m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT,
                                       m->smin[b] + qo * 2 * kQT, P, P <= 1 ? 1 : 0, sub_stride);
                    p0 += P;
                }
                if (sub_stride)
                    hipLaunchKernelGGL(k_gemm_tau_kmin, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, ps, m->smin[b], sub_stride, k, m->tau[b], pairs * 2 * kQT, nq,
                                       m->qnorm[b], m->gmax, e_rel);
                else
                    hipLaunchKernelGGL(k_gemm_tau_min, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, ps, m->smin[b], m->tau[b], pairs * 2 * kQT, nq, m->qnorm[b], m->gmax, e_rel);
            } else {
            const int wpb = kGemmBlock / 64;
            const int sample_wgs = (int)((((int64_t)sample_rows + 31) / 32 + wpb - 1) / wpb);
            const bool streamed = m->mfma16 ? (m->dk16 > kSlabH || m->streamed > 0) : (m->streamed >= 0 ? m->streamed != 0 : m->dk16 > kSlabH);
            const dim3 sgrid(std::min(sample_wgs, grid), pairs);
            if (streamed)
                hipLaunchKernelGGL((k_gemm_proxy_f16<0, 1>), sgrid, dim3(kGemmBlock), kHalfLds, ps, m->gh, m->gnorm, m->qbf[b], m->qinv[b], n, (int64_t)0,
                                   (int64_t)sample_rows, m->dk16, m->tau[b], m->lists[b], m->counts[b], m->sample, sample_rows, 0, 1);
            else
                hipLaunchKernelGGL((k_gemm_proxy_f16<0, 0>), sgrid, dim3(kGemmBlock), kHalfLds, ps, m->gh, m->gnorm, m->qbf[b], m->qinv[b], n, (int64_t)0,
                                   (int64_t)sample_rows, m->dk16, m->tau[b], m->lists[b], m->counts[b], m->sample, sample_rows, 0, 1);
            hipLaunchKernelGGL(k_gemm_tau, dim3(pairs * 2 * kQT), dim3(256), 0, ps, m->sample, (sample_rows + 31) / 32, m->tau[b], nq, m->qnorm[b], m->gmax,
                               e_rel, 2 * kQT, k > 1 ? 1 : 0);
            }
        } else {
            hipLaunchKernelGGL(k_gemm_qnorm, dim3(np * kQT), dim3(64), 0, ps, dq, nq, d, m->qnorm[b]);
            GEMM_HIP(hipMemsetAsync(m->counts[b], 0, (size_t)np * kQT * sizeof(int), ps));
            if (m->precision == FIR_GEMM_F32) {
                hipLaunchKernelGGL(k_gemm_pack_queries, dim3(((kQT / 32) * m->dq8 * 64 + 255) / 256, np), dim3(256), 0, ps, dq, nq, d, m->dq8, m->qm[b]);
                hipLaunchKernelGGL(k_gemm_proxy<0>, dim3(sample_grid, np), dim3(128), lds, ps, m->gm, m->gnorm, m->qm[b], n, (int64_t)0,
                                   (int64_t)sample_rows, m->dq8, m->tau[b], m->lists[b], m->counts[b], m->sample, sample_rows);
            } else {
                hipLaunchKernelGGL(k_gemm_pack_queries_bf16, dim3(((kQT / 32) * m->dk16 * 64 + 255) / 256, np), dim3(256), 0, ps, dq, nq, d, m->dk16,
                                   m->qbf[b]);
                hipLaunchKernelGGL(k_gemm_proxy_bf16<0>, dim3(sample_grid, np), dim3(128), 0, ps, m->gb, m->gnorm, m->qbf[b], n, (int64_t)0,
                                   (int64_t)sample_rows, m->dk16, m->tau[b], m->lists[b], m->counts[b], m->sample, sample_rows);
            }
            hipLaunchKernelGGL(k_gemm_tau, dim3(np * kQT), dim3(256), 0, ps, m->sample, sample_rows, m->tau[b], 0x7FFFFFFF, m->qnorm[b], m->gmax, e_rel, 1,
                               k > 1 ? 1 : 0);
        }
        GEMM_HIP(hipGetLastError());
        GEMM_HIP(hipEventRecord(m->prep_done[b], ps));
        return FIR_OK;
    };
    int rcp = prep(0);
    if (rcp) return rcp;
    for (int sb = 0; sb < nsb; ++sb) {
        const int q0 = sb * sbq;
        const int nq = std::min(sbq, qb - q0);
        const int np = (nq + kQT - 1) / kQT;
        const int b = sb & 1;
        const float* dq = d_queries + (size_t)q0 * qs;
        if (serial_prep) { if (sb >= 1 && (rcp = prep(sb))) return rcp; }
        else if (sb + 1 < nsb && (rcp = prep(sb + 1))) return rcp;
        const bool adaptive = adaptive_for(nq);
        GEMM_HIP(hipStreamWaitEvent(st, m->prep_done[b], 0));
        // ---- the full pass(es) over the gallery: the launch fir_profile_read times and fir_gallery_last_dispatch names ----
        if (m->precision == FIR_GEMM_F16) {
            const int pairs = (np + 1) / 2;
            // (a super-batch of at most 16 / 32 queries: one / two query blocks of the tile are live -- the pass multiplies against those only)
            const bool streamed = m->mfma16 ? (m->dk16 > kSlabH || m->streamed > 0) : (m->streamed >= 0 ? m->streamed != 0 : m->dk16 > kSlabH);
            // (whole, even numbers of the form's units per row: eight pieces streamed, sixteen resident)
            const bool njb_shape = streamed ? !((m->dk16 / kRing) & 1) : (m->dk16 % (4 * kRing)) == 0;
            const int njb = (k == 1 && adaptive && m->mfma16 && !m->dbg_skip && njb_shape && m->few_blocks) ? (nq <= 16 ? 1 : nq <= 32 ? 2 : 8) : 8;
            const int64_t rblocks = (n + 31) / 32;
            // rows longer than the LDS tile (query slabs streamed per unit): 16 readers of one range drift apart, 8 measured better
            const int share_cap = m->share_max > 0 ? (streamed ? std::min(m->share_max, m->share_streamed) : m->share_max) : 16;
            int nlaunch = 0, p_first = 1;
            for (int p0 = 0; p0 < pairs;) {
                int P = 1;
                while (P * 2 <= pairs - p0 && P * 2 <= share_cap) P *= 2;
                if (p0 == 0) p_first = P;
                p0 += P;
                ++nlaunch;
            }
            // algorithmic bytes of ONE launch (what fir_profile_read's event pairs bracket): the fp16 fragments once for the pairs that
            // read them together (once per pair without sharing), every pair's query tile and keys
            auto launch_bytes = [&](int P) { return (m->share_max > 0 ? 1 : P) * ((double)rblocks * m->dk16 * 1024.0) + P * (4.0 * m->dk16 * 1024.0 + 128.0 * 8.0); };
            const double bytes = launch_bytes(p_first);
            const double flops = 2.0 * (double)n * d * 128.0 * p_first;
            int rc2 = FIR_OK;
            bool used_rt = false;
            size_t used_rt_lds = 0;
            // the pairs of the super-batch read the gallery together, a power of two (<= share_max) of them per launch
            for (int p0 = 0; p0 < pairs;) {
                int P = 1;
                while (P * 2 <= pairs - p0 && P * 2 <= share_cap) P *= 2;
                const int share = m->share_max > 0 ? P : 0;
                const dim3 g1 = share > 0 ? dim3(grid, 1) : dim3(grid, P);
                const int nt = (share <= 1) ? 1 : 0;
                const size_t qo = (size_t)p0;
                if ((rc2 = fir_gallery_profile_begin_(m->g, st))) return rc2;
                if (rt_full && share > 0 && grid / 8 >= share) {
                    hipLaunchKernelGGL(rt_main, dim3(grid), dim3(512), rt_lds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT, n, n, 1,
                                       m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT, m->smin[b] + qo * 2 * kQT, share, nt, 0);
                    used_rt = true;
                    used_rt_lds = rt_lds;
                } else if (adaptive)
                    hipLaunchKernelGGL(pick_x(k > 1 ? 4 : 3, streamed, (m->dk16 / kRing) & 1, k > 1 ? 0 : m->dbg_skip, njb), g1, dim3(streamed ? kGemmBlock : m->dbg_block), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64,
                                       m->qinv[b] + qo * 2 * kQT, n, (int64_t)0, n, m->dk16, m->awin[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap,
                                       m->counts[b] + qo * 2 * kQT, m->qnorm[b] + qo * 2 * kQT, sample_rows, share, nt | adapt_dbg | (m->stagger ? 2 : 0) | (m->stagger > 1 ? 32 : 0) | (m->prio ? 16 : 0) | (m->no_block_bound ? 64 : 0), 1,
                                       m->aT[b] + qo * 2 * kQT * (k > 1 ? 8 : 1), k > 1 ? k : 0);
                else if (m->mfma16)
                    hipLaunchKernelGGL(pick_x(1, streamed, (m->dk16 / kRing) & 1), g1, dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64,
                                       m->qinv[b] + qo * 2 * kQT, n, (int64_t)0, n, m->dk16, m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap,
                                       m->counts[b] + qo * 2 * kQT, m->sample, sample_rows, share, nt | (m->stagger ? 2 : 0) | (m->stagger > 1 ? 32 : 0) | (m->prio ? 16 : 0) | (m->no_block_bound ? 64 : 0), 1, (unsigned int*)nullptr, 0);
                else if (streamed)
                    hipLaunchKernelGGL((k_gemm_proxy_f16<1, 1>), g1, dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT,
                                       n, (int64_t)0, n, m->dk16, m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT, m->sample,
                                       sample_rows, share, nt);
                else
                    hipLaunchKernelGGL((k_gemm_proxy_f16<1, 0>), g1, dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT,
                                       n, (int64_t)0, n, m->dk16, m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT, m->sample,
                                       sample_rows, share, nt);
                if ((rc2 = fir_gallery_profile_end_(m->g, st, launch_bytes(P)))) return rc2;
                p0 += P;
            }
            if (used_rt) {
                char nm[64];
                std::snprintf(nm, sizeof nm, "fir::k_gemm_proxy_f16_regtile<%d, false>", m->dk16);
                const void* fp = (const void*)rt_main;
                fir_gallery_note_dispatch_(m->g, fp, nm, sb == 0, grid, nlaunch, 512, used_rt_lds, 128 * p_first, bytes, flops);
            } else
            fir_gallery_note_dispatch_(m->g, m->mfma16 ? (const void*)pick_x(adaptive ? (k > 1 ? 4 : 3) : 1, streamed, (m->dk16 / kRing) & 1, 0, njb)
                                                       : (streamed ? (const void*)k_gemm_proxy_f16<1, 1> : (const void*)k_gemm_proxy_f16<1, 0>),
                                       m->mfma16 ? name_x(streamed, (m->dk16 / kRing) & 1, adaptive, k > 1, njb)
                                                 : (streamed ? "fir::k_gemm_proxy_f16<1, 1>" : "fir::k_gemm_proxy_f16<1, 0>"), sb == 0, grid, m->share_max > 0 ? nlaunch : pairs, kGemmBlock, kHalfLds,
                                       m->share_max > 0 ? 128 * p_first : 128, bytes, flops);
        } else if (m->precision == FIR_GEMM_F32) {
            hipLaunchKernelGGL(k_gemm_proxy<1>, dim3(grid, np), dim3(kGemmBlock), lds, st, m->gm, m->gnorm, m->qm[b], n, (int64_t)0, n, m->dq8,
                               m->tau[b], m->lists[b], m->counts[b], m->sample, sample_rows);
            fir_gallery_note_dispatch_(m->g, (const void*)k_gemm_proxy<1>, "fir::k_gemm_proxy<1>", sb == 0, grid, np, kGemmBlock, lds, 64,
                                       np * ((double)((n + 31) / 32) * m->dq8 * 1024.0), 2.0 * (double)n * d * 64.0 * np);
        } else {
            // pairs of passes share one read of the gallery (128 queries per wave); an odd last pass goes alone
            const int pairs = m->wide ? np / 2 : 0;
            if (pairs > 0)
                hipLaunchKernelGGL(k_gemm_proxy_bf16_wide, dim3(grid, pairs), dim3(kGemmBlock), kWideLds, st, m->gb, m->gnorm, m->qbf[b], n, m->dk16,
                                   m->tau[b], m->lists[b], m->counts[b]);
            if (np > 2 * pairs) {
                const size_t p0 = (size_t)2 * pairs;
                hipLaunchKernelGGL(k_gemm_proxy_bf16<1>, dim3(grid, np - 2 * pairs), dim3(kGemmBlock), lds, st, m->gb, m->gnorm,
                                   m->qbf[b] + p0 * (kQT / 32) * m->dk16 * 128, n, (int64_t)0, n, m->dk16, m->tau[b] + p0 * kQT,
                                   m->lists[b] + p0 * kQT * kListCap, m->counts[b] + p0 * kQT, m->sample + p0 * kQT * sample_rows, sample_rows);
            }
            fir_gallery_note_dispatch_(m->g, (const void*)k_gemm_proxy_bf16_wide, pairs > 0 ? "fir::k_gemm_proxy_bf16_wide" : "fir::k_gemm_proxy_bf16<1>", sb == 0,
                                       grid, pairs > 0 ? pairs : np, kGemmBlock, pairs > 0 ? (size_t)kWideLds : lds, pairs > 0 ? 128 : 64,
                                       (pairs + (np - 2 * pairs)) * ((double)((n + 31) / 32) * m->dk16 * 2048.0), 2.0 * (double)n * d * 64.0 * np);
        }
        if (adaptive) {
            const int pairs_a = (np + 1) / 2;
            hipLaunchKernelGGL(k_gemm_adapt_final, dim3((pairs_a * 2 * kQT + 255) / 256), dim3(256), 0, st, m->aT[b], m->qnorm[b], m->tau[b], pairs_a * 2 * kQT, nq, k > 1 ? k : 0);
        }
        GEMM_HIP(hipEventRecord(m->main_done[b], st));
        // exact re-rank + certificate of this super-batch on the side stream, under the next one's full pass
        hipStream_t rs = one_stream ? st : m->side;
        if (!one_stream) GEMM_HIP(hipStreamWaitEvent(m->side, m->main_done[b], 0));
        const RerankFb fb = {m->fb_state, m->fb_list, m->fb_tau2, q0, nullptr, 0};     // uncertified queries go on the call's list
        if (k == 1)
            hipLaunchKernelGGL(k_gemm_rerank, dim3(nq), dim3(64), (size_t)(m->rerank_group + 1) * (m->dp4 + 1) * sizeof(float4), rs, m->lists[b], m->counts[b],
                               m->tau[b], m->gal4, dq, m->qnorm[b], m->gmax, n, d, m->dp4, m->v.row_offset, e_rel, m->rerank_group,
                               (unsigned long long*)d_keys + q0, m->ok + q0, qs, m->rowmajor, fb);
        else
            hipLaunchKernelGGL(k_gemm_rerank_topk, dim3(nq), dim3(64), (size_t)(m->rerank_group + 1) * (m->dp4 + 1) * sizeof(float4), rs, m->lists[b],
                               m->counts[b], m->tau[b], m->gal4, dq, m->qnorm[b], m->gmax, n, d, m->dp4, m->v.row_offset, e_rel, m->rerank_group, k,
                               (unsigned long long*)d_keys + (size_t)q0 * k, m->ok + q0, qs, m->rowmajor, fb);
        GEMM_HIP(hipEventRecord(m->rerank_done[b], rs));
        m->passes += np;
    }
    if (!one_stream) GEMM_HIP(hipStreamWaitEvent(st, m->rerank_done[(nsb - 1) & 1], 0));   // join the side stream (it is in order: the last re-rank is the last thing on it)
    GEMM_HIP(hipGetLastError());
#ifdef FIR_AUDIT
    if (const char* path = fir_knob_("FIR_GEMM_DUMP_PHASES")) {     // audit builds, with FIR_GEMM_DBG_SKIP=256: the load-burst timestamps of the last super-batch's first pair
        GEMM_HIP(hipStreamSynchronize(st));
        std::vector<unsigned long long> h(8 * 256);
        GEMM_HIP(hipMemcpy(h.data(), m->lists[(nsb - 1) & 1] + (size_t)(2 * kQT - 1) * kListCap + 2048, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = std::fopen(path, "w")) {
            for (int w = 0; w < 8; ++w)
                for (int u = 0; u < 240; ++u) std::fprintf(f, "%d %d %llu %llu\n", w, u, h[(size_t)w * 256 + u] >> 20, h[(size_t)w * 256 + u] & 0xFFFFFull);
            std::fclose(f);
        }
    }
    if (fir_knob_("FIR_GEMM_DEBUG_COUNTS")) {       // audit builds: appended rows per query of the last super-batch (synchronises)
        GEMM_HIP(hipStreamSynchronize(st));
        const int nql = std::min(sbq, qb - (nsb - 1) * sbq);
        std::vector<int> hc((size_t)nql), h_ok((size_t)nql);
        GEMM_HIP(hipMemcpy(hc.data(), m->counts[(nsb - 1) & 1], (size_t)nql * sizeof(int), hipMemcpyDeviceToHost));
        GEMM_HIP(hipMemcpy(h_ok.data(), m->ok + (size_t)(nsb - 1) * sbq, (size_t)nql * sizeof(int), hipMemcpyDeviceToHost));
        long long sum = 0;
        int mx = 0, bad = 0;
        for (int v : hc) { sum += v; mx = std::max(mx, v); }
        for (int v : h_ok) bad += v ? 0 : 1;
        std::vector<float> ht((size_t)nql);
        GEMM_HIP(hipMemcpy(ht.data(), m->tau[(nsb - 1) & 1], (size_t)nql * sizeof(float), hipMemcpyDeviceToHost));
        double ts = 0;
        int ninf = 0;
        for (float v : ht) { if (v < 1e30f) ts += v; else ++ninf; }
        std::fprintf(stderr, "fir_gemm: appended rows per query (last super-batch of %d): mean %.1f, max %d; tau: mean %.6f, %d not finite; %d uncertified\n", nql,
                     (double)sum / nql, mx, ts / std::max(1, nql - ninf), ninf, bad);
    }
#endif
    // uncertified queries: a second matrix-core pass with the tightest bound the first one can justify, then the exact device scan
    // (calls of at most 32 queries skip the second-chance rounds: the exact device scan reads the gallery once per eight queries, about what
    // one more matrix-core pass costs, and four launches fewer are 20 us of such a call)
    const int sc_rounds = x_flow && qb > 32 ? std::min(kScRounds, (qb + kScQueries - 1) / kScQueries) : 0;
    return gemm_finish_(m, d_queries, k, d_keys, st, e_rel, sc_rounds);
}

extern "C" {

int fir_gemm_search_top1_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, uint64_t* d_keys, void* stream) {
    return gemm_search(m, d_queries, qb, 1, d_keys, stream);
}

int fir_gemm_search_topk_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, int32_t k, uint64_t* d_keys, void* stream) {
    return gemm_search(m, d_queries, qb, k, d_keys, stream);
}

// 1..8 queries against a gallery too large for the caches: one pass over the fp16 copy (k_gemm_scan_f16), the rows within one
// window of the smallest proxy of ALL rows, exact re-rank, certificate; everything on `stream`, one synchronisation.
int fir_gemm_search_few_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, uint64_t* d_keys, void* stream) {
    if (!m || !d_keys || !d_queries) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 1 || qb > 8) return gemm_fail(FIR_ERR_ARG, "qb=%d outside [1,8]", qb);
    if (m->precision != FIR_GEMM_F16) return gemm_fail(FIR_ERR_ARG, "the few-query form needs the fp16 copy");
    GEMM_HIP(hipSetDevice(m->v.device));
    hipStream_t st = stream ? (hipStream_t)stream : m->v.stream;
    const int d = m->feat, qs = m->v.d;
    const int64_t n = m->v.n;
    if (n == 0) return fir_search_top1_exact_keys_dev_(m->g, d_queries, qb, 0, d, d_keys, st);
    const int nqt = qb <= 1 ? 1 : qb <= 2 ? 2 : qb <= 4 ? 4 : 8;
    if (nqt > m->proxies_nq) {
        if (m->proxies) GEMM_HIP(hipFree(m->proxies));
        m->proxies = nullptr;
        m->proxies_nq = 0;
        GEMM_HIP(hipMalloc((void**)&m->proxies, (size_t)nqt * n * sizeof(float)));     // (FIR_ERR_NOMEM: the caller takes the exact scan)
        m->proxies_nq = nqt;
    }
    if (m->lists_cap < 2 * kQT) {
        for (int bb = 0; bb < 2; ++bb) {
            (void)hipFree(m->lists[bb]); (void)hipFree(m->counts[bb]);
            m->lists[bb] = nullptr; m->counts[bb] = nullptr;
        }
        m->lists_cap = 0;
        for (int bb = 0; bb < 2; ++bb) {
            GEMM_HIP(hipMalloc((void**)&m->lists[bb], (size_t)2 * kQT * kListCap * sizeof(unsigned long long)));
            GEMM_HIP(hipMalloc((void**)&m->counts[bb], (size_t)2 * kQT * sizeof(int)));
        }
        m->lists_cap = 2 * kQT;
    }
    {
        const int rcr = gemm_fb_reserve_(m, qb);
        if (rcr) return rcr;
    }
    const float e_rel = m->erel_scale * (8.0f * (float)d * 5.9604645e-8f + 9.765625e-4f * 1.0625f);     // as gemm_search (one fp16 term)
    const int b = 0;
    hipLaunchKernelGGL(k_gemm_qprep_f16, dim3(2 * kQT), dim3(64), 0, st, d_queries, qb, d, m->gallery_exp, m->qnorm[b], m->qmul[b], m->qinv[b], qs, m->counts[b],
                       (float*)nullptr, (unsigned int*)nullptr, (const float*)nullptr, 0.f, 0, m->smin[b], m->fb_state);
    if (m->mfma16) hipLaunchKernelGGL(k_gemm_pack_queries_f16x, dim3((4 * m->dk16 * 64 + 255) / 256, 1), dim3(256), 0, st, d_queries, qb, d, m->dk16, m->qmul[b], m->qbf[b], qs);
    else hipLaunchKernelGGL(k_gemm_pack_queries_f16, dim3((4 * m->dk16 * 64 + 255) / 256, 1), dim3(256), 0, st, d_queries, qb, d, m->dk16, m->qmul[b], m->qbf[b], qs);
    const dim3 grid((unsigned)std::min<int64_t>((int64_t)m->v.cus * 8, ((n + 31) / 32 + 3) / 4));
    const size_t lds = (size_t)m->dk16 * 2 * nqt * sizeof(uint4);
#define FIR_FEW(NQ) do { if (m->mfma16) hipLaunchKernelGGL(k_gemm_scan_f16x<NQ>, grid, dim3(256), lds, st, m->gh, m->gnorm, m->qbf[b], m->qinv[b], n, m->dk16, m->proxies, m->smin[b]); \
                         else hipLaunchKernelGGL(k_gemm_scan_f16<NQ>, grid, dim3(256), lds, st, m->gh, m->gnorm, m->qbf[b], m->qinv[b], n, m->dk16, m->proxies, m->smin[b]); } while (0)
    {
        int rcp = fir_gallery_profile_begin_(m->g, st);              // (no-ops unless profiling is on: a profiled run takes the same path as an unprofiled one)
        if (rcp) return rcp;
        if (nqt == 1) FIR_FEW(1); else if (nqt == 2) FIR_FEW(2); else if (nqt == 4) FIR_FEW(4); else FIR_FEW(8);
        if ((rcp = fir_gallery_profile_end_(m->g, st, (double)((n + 31) / 32) * m->dk16 * 1024.0 + (double)nqt * n * 4.0))) return rcp;
    }
#undef FIR_FEW
    fir_gallery_note_dispatch_(m->g, nqt == 1 ? (const void*)k_gemm_scan_f16<1> : nqt == 2 ? (const void*)k_gemm_scan_f16<2> : nqt == 4 ? (const void*)k_gemm_scan_f16<4>
                                                                                                                                             : (const void*)k_gemm_scan_f16<8>,
                               "fir::k_gemm_scan_f16", 1, (int)grid.x, 1, 256, lds, nqt, (double)((n + 31) / 32) * m->dk16 * 1024.0 + (double)nqt * n * 4.0,
                               2.0 * (double)n * d * nqt);
    hipLaunchKernelGGL(k_gemm_tau_min, dim3(1), dim3(256), 0, st, m->smin[b], m->tau[b], 2 * kQT, qb, m->qnorm[b], m->gmax, e_rel);
    hipLaunchKernelGGL(k_gemm_select, dim3((unsigned)std::min<int64_t>(1024, (n + 255) / 256), qb), dim3(256), 0, st, m->proxies, n, m->tau[b], m->lists[b], m->counts[b]);
    const RerankFb fb = {m->fb_state, m->fb_list, m->fb_tau2, 0, nullptr, 0};
    hipLaunchKernelGGL(k_gemm_rerank, dim3(qb), dim3(64), (size_t)(m->rerank_group + 1) * (m->dp4 + 1) * sizeof(float4), st, m->lists[b], m->counts[b], m->tau[b], m->gal4,
                       d_queries, m->qnorm[b], m->gmax, n, d, m->dp4, m->v.row_offset, e_rel, m->rerank_group, (unsigned long long*)d_keys, m->ok, qs, m->rowmajor, fb);
    GEMM_HIP(hipGetLastError());
    m->passes += 1;
    // (the bound of this form already is the smallest proxy of ALL rows + one window: a second pass could not do better; what is
    // not certified -- NaN operands, thousands of ties -- goes to the exact device scan, in stream order)
    return gemm_finish_(m, d_queries, 1, d_keys, st, e_rel, 0);
}

// Host-pointer form for fir_search_top1 / fir_search_topk: h_queries -> d_stage (qb rows of the gallery's length) super-batch by
// super-batch, under the full passes of the one before; keys as the device-pointer forms.
int fir_gemm_search_staged_(fir_gemm* m, const float* h_queries, float* d_stage, int32_t qb, int32_t k, uint64_t* d_keys, void* stream) {
    if (!h_queries || !d_stage) return gemm_fail(FIR_ERR_ARG, "NULL argument");
    return gemm_search(m, d_stage, qb, k, d_keys, stream, h_queries);
}

}  // extern "C"

#include "fir_gemm_f64.h"
