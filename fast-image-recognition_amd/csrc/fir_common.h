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

enum { kL2 = 0, kChi2 = 1, kKL = 2 };
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
    } else {
        const float s = l + r;                                  // db_features.cpp:29,33-36
        float a = acc;
        if (s > 0.0f) {
            if (l > 0.0f) a = a + l * logf(2.0f * l / s);
            if (r > 0.0f) a = a + r * logf(2.0f * r / s);
        }
        return a;
    }
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
