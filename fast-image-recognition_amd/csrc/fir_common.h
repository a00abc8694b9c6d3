// fir_common.h -- device helpers shared by the library's translation units: key packing, the
// reference's per-feature arithmetic (accum), wave reductions.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace fir {

constexpr int kTileRows = 64;   // rows per tile == lanes per wavefront (gfx950 wave64)
constexpr int kBlock = 256;     // 4 waves per workgroup, one per SIMD
constexpr float kNotFound = 100000.0f;  // db_features.cpp:323, ann.cpp:116
constexpr uint64_t kKeyNone = 0xFFFFFFFFFFFFFFFFull;

enum { kL2 = 0, kChi2 = 1, kKL = 2,
       kChi2InRange = 3, kKLInRange = 4,     // kernel-internal: the same arithmetic for operands known to be 0 or in [2^-26, 2^16]
       kChi2Approx = 5,                      // kernel-internal: chi-square with a 1-ulp reciprocal, NOMINATES rows only (fir_capi.hip: topk_lists_dev)
       kChi2Harm = 6,                        // kernel-internal: chi-square as sum(l) + sum(r) - 4 sum 1/(1/l + 1/r): the scan adds up the harmonic terms, NOMINATES only
       kKLEnt = 7 };                         // kernel-internal: KL as ln2 (sum(l log2 l + l) + sum(r log2 r + r) - sum (l + r) log2 (l + r)): the scan adds up the last sum, NOMINATES only
enum { kEpiTop1 = 0, kEpiTopK = 1, kEpiStore = 2, kEpiAppend = 3 };

typedef const float __attribute__((address_space(4)))* sfloat_p;  // constant AS => s_load when uniform

// float bits -> uint32 whose unsigned order equals the float order (negatives included).
__host__ __device__ __forceinline__ uint32_t f32_orderable(float f) {
    uint32_t b;
    __builtin_memcpy(&b, &f, 4);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__host__ __device__ __forceinline__ float f32_from_orderable(uint32_t o) {
    uint32_t b = (o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o;
    float f;
    __builtin_memcpy(&f, &b, 4);
    return f;
}
__host__ __device__ __forceinline__ uint64_t key_pack(float dist, uint32_t idx) {
    return ((uint64_t)f32_orderable(dist + 0.0f) << 32) | (uint64_t)idx;  // +0.0f: -0 -> +0
}

// ---- chi-square / KL for operands in the "plain" range ---------------------------------------------------------------
// The correctly rounded f32 division the compiler emits is v_div_scale x2, v_rcp, seven fma/mul steps, v_div_fmas and
// v_div_fixup. For a denominator and a numerator that are far from the ends of the exponent range the two v_div_scale
// return their operand unchanged with VCC = 0, v_div_fmas is then a plain fma and v_div_fixup passes the quotient
// through: what remains is the reciprocal with one Newton step and the seven arithmetic steps below -- the same
// instructions on the same operands, so the same bits. in_plain_range() is the condition the gallery upload and the
// query transposition check for every value (0, or 2^-26 <= x <= 2^16: then x + y is 0 or in [2^-26, 2^17], (x - y)^2 is 0
// or in [2^-98, 2^32] -- above the 2^-103 where v_div_scale starts scaling numerators --, every quotient below lies in
// [2^-115, 2^58], and none of the v_div_scale / v_div_fixup cases applies; L1-normalised 1536-feature rows, whose
// smallest non-zero entries are near 1e-7, are inside);
// one value outside it and the kernels take accum<kChi2> / accum<kKL> for the whole call. Halves the VALU work per
// element of the chi-square scan (packs pairwise into v_pk_fma_f32) and more for KL, where the two quotients also share
// the reciprocal and the divergent `if (l > 0)` branches become a v_max.
__host__ __device__ __forceinline__ bool in_plain_range(float x) {
    uint32_t b;
    __builtin_memcpy(&b, &x, 4);
    return b == 0u || (b >= 0x32800000u && b <= 0x47800000u);    // +0, or 2^-26 .. 2^16 (negatives, NaN, inf: no)
}
#ifdef __HIP_DEVICE_COMPILE__
__device__ __forceinline__ float rcp_newton(float s) {             // first three steps of the f32 division sequence
    const float y0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float div_plain(float n, float s, float y) {   // the remaining five; y = rcp_newton(s)
    float q = n * y;
    float r = __builtin_fmaf(-s, q, n);
    q = __builtin_fmaf(r, y, q);
    r = __builtin_fmaf(-s, q, n);
    return __builtin_fmaf(r, y, q);
}
__device__ __forceinline__ float log_plain(float x) {              // logf(x) as the compiler expands it, for normal finite x > 0
    const float y = __builtin_amdgcn_logf(x);                      // v_log_f32 (log2)
    const float p = y * 0x1.62e42ep-1f;
    float pl = __builtin_fmaf(y, 0x1.62e42ep-1f, -p);
    pl = __builtin_fmaf(y, 0x1.efa39ep-25f, pl);
    return p + pl;
}
#endif

// One feature of the reference's distance loop, lhs = query (test image), rhs = gallery row
// (ImageInfo::distance, db_features.h:24-26). The translation unit is compiled with
// -ffp-contract=off: sub, mul, add (and the chi-square divide) each round once, like the
// reference's SSE scalar code.
template <int METRIC>
__device__ __forceinline__ float accum(float acc, float l, float r) {
    if constexpr (METRIC == kL2) {
        const float df = l - r;
        return acc + df * df;                                   // db_features.cpp:26
    } else if constexpr (METRIC == kChi2) {
        const float s = l + r;
        const float df = l - r;
        const float term = df * df / s;                         // db_features.cpp:31
        return (s > 0.0f) ? acc + term : acc;                   // db_features.cpp:29
    } else if constexpr (METRIC == kKL) {
        const float s = l + r;                                  // db_features.cpp:29,33-36
        float a = acc;
        if (s > 0.0f) {
            if (l > 0.0f) a = a + l * logf(2.0f * l / s);
            if (r > 0.0f) a = a + r * logf(2.0f * r / s);
        }
        return a;
    }
#ifdef __HIP_DEVICE_COMPILE__
    else if constexpr (METRIC == kChi2Approx) {
        // (l - r)^2 * rcp(l + r): v_rcp_f32 is good to 1 ulp and the product rounds once more, so every term is within
        // 2^-22 of the real quotient (the exact sequence's: 2^-24); all terms are >= 0, so whatever the order of the adds the sum
        // is within (2d + 8) 2^-24 of the reference's. 5.5 issue slots per element instead of 11. Plain-range operands only.
        const float df = l - r;
        const float s = __builtin_fmaxf(l + r, 0x1p-60f);        // l + r == 0 only for l == r == 0: then the term is 0 * 2^60 = +0
        return acc + (df * df) * __builtin_amdgcn_rcpf(s);
    }
    else if constexpr (METRIC == kChi2Harm) {
        // l = 1 / (query value) (2^60 for 0), r = gallery value: one harmonic term l_k r_k / (l_k + r_k) = 1 / (1/l_k + 1/r_k)
        // (range edges only: whole chunks take TileAcc::chunk's two-terms-per-reciprocal form)
        return acc + __builtin_amdgcn_rcpf(l + __builtin_amdgcn_rcpf(r));
    }
    else if constexpr (METRIC == kKLEnt) {
        // l = query value + 2^-100 (k_transpose_queries: l + r > 0 also where both are 0, and the term is then -100 * 2^-100),
        // r = gallery value: one term (l + r) log2 (l + r) of the entropy form, v_log_f32 is the base-2 logarithm
        const float s = l + r;
        return __builtin_fmaf(s, __builtin_amdgcn_logf(s), acc);
    }
    else if constexpr (METRIC == kChi2InRange) {
        const float df = l - r;
        const float n = df * df;
        const float s = __builtin_fmaxf(l + r, 0x1p-30f);       // l + r == 0 only for l == r == 0: then n == 0 and the term is +0
        return acc + div_plain(n, s, rcp_newton(s));
    } else {
        // the two quotients 2l/s and 2r/s share the reciprocal and go through the division steps and the log's
        // multiply as ONE packed pair (v_pk_mul_f32 / v_pk_fma_f32): the same operations per component as div_plain / log_plain
        typedef float f2v __attribute__((ext_vector_type(2)));
        const float s = __builtin_fmaxf(l + r, 0x1p-30f);
        const float y = rcp_newton(s);
        const f2v lr = {l, r}, ys = {y, y}, ns = {-s, -s};
        const f2v n2 = lr * 2.0f;
        f2v q2 = n2 * ys;
        f2v r2 = __builtin_elementwise_fma(ns, q2, n2);
        q2 = __builtin_elementwise_fma(r2, ys, q2);
        r2 = __builtin_elementwise_fma(ns, q2, n2);
        q2 = __builtin_elementwise_fma(r2, ys, q2);
        // l == 0: the quotient is 0, clamped; l * log(clamp) = -0 and acc + -0 = acc, as if the term had been skipped
        const f2v lg = {__builtin_amdgcn_logf(__builtin_fmaxf(q2.x, 0x1p-60f)), __builtin_amdgcn_logf(__builtin_fmaxf(q2.y, 0x1p-60f))};
        const f2v c = {0x1.62e42ep-1f, 0x1.62e42ep-1f}, cc = {0x1.efa39ep-25f, 0x1.efa39ep-25f};
        const f2v p = lg * c;
        f2v pl = __builtin_elementwise_fma(lg, c, -p);
        pl = __builtin_elementwise_fma(lg, cc, pl);
        const f2v t = lr * (p + pl);
        const float a = acc + t.x;
        return a + t.y;
    }
#else
    else return acc;
#endif
}

// 64-bit wave-wide minimum (all lanes end with the result).
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint64_t o = __shfl_xor((unsigned long long)v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}


}  // namespace fir
