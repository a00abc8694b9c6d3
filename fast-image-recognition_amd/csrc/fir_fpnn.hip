// fir_fpnn.hip -- FPNNClassifier (orthogonal-series / trigonometric PNN) of qt_cpp/classification.cpp:618-791 on gfx950.
//
// The model is dense: per (feature f, class c) 2J+1 Fourier coefficients (classification.cpp:676-693). Training is a
// sum over each class's training rows, prediction a D x C x J contraction per query followed by a fast log2 and a
// float sum over the features. Both are evaluated in the reference's order:
//   k_fpnn_train   one thread per (class, harmonic j, feature): a_cos/a_sin += cos/sin(PI (j+1) val) * cur_mult * (J-j) / (J (J+1))
//                  over the class's rows in training-set order (:680-690). Lanes run over the features, so the training
//                  rows (row-major doubles) are read coalesced.
//   k_fpnn_terms   a workgroup per (few features, query): val = normalize(x) (:647-655), cos/sin(PI val) and the
//                  angle-addition recurrence of :706-711 for the J harmonics, then one thread per (feature, class):
//                  probab = a0 + sum_j (a_cos*cos_j + a_sin*sin_j) in double (:717-721) -> fasterlog2((float)probab).
//                  The terms are independent, so a ONE-query call is spread over d / F workgroups (116 -> see
//                  profiles/r02_latency_secondary.txt us per call at 3030 x 256, 101 classes).
//   k_fpnn_predict one workgroup per query, threads over the classes: outputs[c] += term in float, features in order
//                  (:722). SEQ form (:736-791): the same in 32-feature chunks with the class pruning rule after each.
// The model is kept transposed -- at[(f*(2J+1) + k) * C + c] -- so that the classes of a workgroup read consecutive
// doubles; fir_fpnn_get_model returns the reference's a[(f*C + c)*(2J+1) + k].
// The translation unit is built with -ffp-contract=off. cos/sin are the device's double-precision functions (<= 2 ulp);
// the reference's come from glibc. A last-bit difference there survives into an output only when it flips the rounding
// of the double -> float conversion in front of fasterlog2 (about one evaluation in 1e8), so outputs are compared with a
// small tolerance and classes exactly.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_internal.h"

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 32;        // PNNClassifier::delta_features_count, classification.cpp:182
constexpr int kMaxJ = 64;
constexpr int kPredBatch = 64;    // queries per launch (bounds the term scratch)

thread_local char g_fpnn_err[512];
int fpnn_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_fpnn_err, sizeof(g_fpnn_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_fpnn_err);
    return code;
}
#define FPNN_HIP(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fpnn_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP,       \
                                               "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// FPNNClassifier::normalize, classification.cpp:637-659 (the active `#elif 1` arm + the clamp to [-0.5, 0.5]).
__device__ __forceinline__ double fpnn_normalize(double x, double avg, double sd, double scale) {
    double val = (sd != 0) ? scale * (x - avg) / sd : 0;
    const double max_val = 0.5;
    if (val < -max_val) val = -max_val;
    else if (val > max_val) val = max_val;
    return val;
}

// fasterlog2, classification.cpp:64-73.
__device__ __forceinline__ float fasterlog2(float x) {
    const uint32_t vi = __float_as_uint(x);
    const float mx = __uint_as_float((vi & 0x007FFFFFu) | (0x7eu << 23));
    float y = (float)vi;
    y = (float)((double)y * (1.0 / (1 << 23)));
    return y - 124.22544637f - 1.498030302f * mx - 1.72587999f / (0.3520887068f + mx);
}

__global__ void __launch_bounds__(kBlock) k_fpnn_train(const double* __restrict__ rows, int d, const int32_t* __restrict__ class_off, int C, int J,
                                                       const double* __restrict__ avg, const double* __restrict__ sd, double scale,
                                                       double* __restrict__ at) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= (int64_t)C * J * d) return;
    const int f = (int)(o % d);
    const int j = (int)((o / d) % J);
    const int c = (int)(o / ((int64_t)d * J));
    const double PI = 3.141592653589793;                     // atan(1.0)*4, classification.cpp:659
    const int t0 = class_off[c], t1 = class_off[c + 1];
    const double cur_mult = 1.0 / (double)(t1 - t0);         // :679
    const double a = avg[f], s = sd[f];
    const double wj = (double)(J - j), den = (double)((int64_t)J * (J + 1));
    const double freq = PI * (double)(j + 1);
    double ac = 0.0, as = 0.0;
    for (int t = t0; t < t1; ++t) {
        const double val = fpnn_normalize(rows[(int64_t)t * d + f], a, s, scale);
        ac += cos(freq * val) * cur_mult * wj / den;         // :684
        as += sin(freq * val) * cur_mult * wj / den;         // :685
    }
    const int K = 2 * J + 1;
    at[((int64_t)f * K + 2 * j + 1) * C + c] = ac;
    at[((int64_t)f * K + 2 * j + 2) * C + c] = as;
    if (j == 0) at[((int64_t)f * K) * C + c] = 0.5;           // :678
}

// terms[(q*d + f)*C + c] = fasterlog2((float)probab(q, f, c)), classification.cpp:706-721: every (query, feature, class)
// term is independent of the others, so they are spread over ceil(d / F) x nq workgroups (a one-query call keeps the
// whole chip busy instead of one CU); only the per-class float sums over the features are ordered, and k_fpnn_predict
// does those. The first F threads run the angle-addition recurrences (:708-711) of the workgroup's F features into LDS.
__global__ void __launch_bounds__(kBlock) k_fpnn_terms(const double* __restrict__ q, int d, int C, int J, int F, const double* __restrict__ avg,
                                                       const double* __restrict__ sd, double scale, const double* __restrict__ at,
                                                       float* __restrict__ terms) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_t[];
    double* trig = (double*)smem_t;                           // [F][2J]: cos_vals, sin_vals
    const int qi = blockIdx.y, f0 = blockIdx.x * F;
    const int nf = min(F, d - f0);
    const int K = 2 * J + 1;
    if ((int)threadIdx.x < nf) {
        const int f = f0 + threadIdx.x;
        const double PI = 3.141592653589793;
        const double val = fpnn_normalize(q[(int64_t)qi * d + f], avg[f], sd[f], scale);
        double* cv = trig + (size_t)threadIdx.x * 2 * J;
        double* sv = cv + J;
        const double c0 = cos(PI * val), s0 = sin(PI * val);
        double cp = c0, sp = s0;
        cv[0] = c0;
        sv[0] = s0;
        for (int j = 1; j < J; ++j) {                         // :708-711
            const double cn = cp * c0 - sp * s0;
            const double sn = cp * s0 + sp * c0;
            cv[j] = cn;
            sv[j] = sn;
            cp = cn;
            sp = sn;
        }
    }
    __syncthreads();
    for (int it = threadIdx.x; it < nf * C; it += kBlock) {
        const int fl = it / C, c = it - fl * C;
        const double* cv = trig + (size_t)fl * 2 * J;
        const double* sv = cv + J;
        const double* __restrict__ m = at + (int64_t)(f0 + fl) * K * C + c;
        double probab = m[0];
        for (int j = 0; j < J; ++j)
            probab += (m[(int64_t)(2 * j + 1) * C] * cv[j] + m[(int64_t)(2 * j + 2) * C] * sv[j]);   // :719 / :763
        terms[((int64_t)qi * d + f0) * C + it] = fasterlog2((float)probab);
    }
}

struct BestF {
    float v;
    int i;
};
// first maximum in class order among candidates (v desc, i asc); i = -1 when nobody beat -FLT_MAX
__device__ BestF block_first_max(float v, int i, BestF* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(i, off, 64);
        if (ov > v || (ov == v && (unsigned)oi < (unsigned)i)) { v = ov; i = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = BestF{v, i};
    __syncthreads();
    BestF r = red[0];
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w)
        if (red[w].v > r.v || (red[w].v == r.v && (unsigned)red[w].i < (unsigned)r.i)) r = red[w];
    return r;
}
__device__ int block_sum(int v, int* red) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < kBlock / 64; ++w) r += red[w];
    return r;
}

// One workgroup per query over the terms k_fpnn_terms left. Dynamic LDS: C floats (outputs) + nsub x C floats (terms of nsub
// features) + C bytes (classes_to_check).
template <bool SEQ>
__global__ void __launch_bounds__(kBlock) k_fpnn_predict(int d, int C, const float* __restrict__ terms, float output_ratio, int nsub, int32_t* __restrict__ best_class,
                                                         float* __restrict__ outputs_out, int32_t* __restrict__ chunks_out,
                                                         unsigned long long* ticket_word, unsigned long long ticket) {
    // ticket_word (one-query calls only: a single workgroup): once every thread's results are visible to the host, thread 0
    // publishes the call's ticket there and the host spins on that word instead of synchronising the stream.
    extern __shared__ unsigned char smem[];
    float* outputs = (float*)smem;
    float* lg = outputs + C;                                  // nsub x C fast-log terms
    unsigned char* alive = smem + (size_t)(1 + nsub) * C * sizeof(float);
    __shared__ BestF redb[kBlock / 64];
    __shared__ int redi[kBlock / 64];
    const int q = blockIdx.x;
    const float output_delta = fasterlog2(output_ratio);     // the constructor's fastlog(output_ratio), :621
    for (int c = threadIdx.x; c < C; c += kBlock) { outputs[c] = 0.0f; alive[c] = 1; }
    __syncthreads();
    int bestClass = -1, chunks = 0;
    for (int cur = 0; cur < d; cur += SEQ ? kChunk : d) {
        const int max_fi = SEQ ? min(cur + kChunk, d) : d;
        ++chunks;
        // `nsub` features at a time: the whole workgroup stages their terms in LDS (independent, coalesced loads), then one
        // thread per class adds that class's terms in feature order; a class no longer checked keeps its sum (:758).
        for (int f0 = cur; f0 < max_fi; f0 += nsub) {
            const int nf = min(nsub, max_fi - f0);
            const float* __restrict__ src = terms + ((int64_t)q * d + f0) * C;
            for (int it = threadIdx.x; it < nf * C; it += kBlock) lg[it] = src[it];
            __syncthreads();
            for (int c = threadIdx.x; c < C; c += kBlock) {
                if (!alive[c]) continue;
                float out = outputs[c];
                for (int fl = 0; fl < nf; ++fl) out += lg[fl * C + c];   // :722 / :765
                outputs[c] = out;
            }
            __syncthreads();
        }
        // first maximum among the classes still checked, strict '<' from -FLT_MAX (:726-733 / :770-776)
        float bv = -FLT_MAX;
        int bi = -1;
        for (int c = threadIdx.x; c < C; c += kBlock)
            if (alive[c] && bv < outputs[c]) { bv = outputs[c]; bi = c; }
        const BestF b = block_first_max(bv, bi, redb);
        if (b.i >= 0) bestClass = b.i;
        if (!SEQ) break;
        const float output_threshold = b.v + output_delta * (float)max_fi;   // :778 (size_t -> float)
        int variants = 0;
        for (int c = threadIdx.x; c < C; c += kBlock) {      // :780-785: every class is tested, dropped ones are never revived
            if (outputs[c] < output_threshold) alive[c] = 0;
            else ++variants;
        }
        variants = block_sum(variants, redi);
        if (variants == 1) break;                             // :786 (uniform: every thread holds the block sum)
    }
    if (threadIdx.x == 0) {
        best_class[q] = bestClass;
        if (chunks_out) chunks_out[q] = chunks;
    }
    if (outputs_out)
        for (int c = threadIdx.x; c < C; c += kBlock) outputs_out[(int64_t)q * C + c] = outputs[c];
    if (ticket_word) {
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(ticket_word, ticket, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// at[(f*K + k)*C + c] -> a[(f*C + c)*K + k]
__global__ void __launch_bounds__(kBlock) k_fpnn_untranspose(const double* __restrict__ at, int d, int C, int K, double* __restrict__ a) {
    const int64_t o = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (o >= (int64_t)d * C * K) return;
    const int k = (int)(o % K);
    const int c = (int)((o / K) % C);
    const int f = (int)(o / ((int64_t)K * C));
    a[o] = at[((int64_t)f * K + k) * C + c];
}

struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 16)); }
    template <typename T> T* as() { return (T*)p; }
};

}  // namespace

struct fir_fpnn {
    int device = 0;
    int d = 0, C = 0, J = 0;
    double scale = 1.0;
    hipStream_t stream = nullptr;
    Buf at, avg, sd, terms;
    void* pin = nullptr;        // pinned staging: queries in, outputs / classes / chunk counts out, then the ticket word
    unsigned long long ticket = 0;    // one-query calls so far
};

namespace {
// Queries go in and verdicts come out through pinned, device-visible host memory: the kernels read / write it directly,
// so a call is two launches and ONE synchronisation per batch, with no copy engine in between.
int predict_common(fir_fpnn* h, const double* queries, int32_t qb, bool seq, float output_ratio, int32_t* best_class, float* outputs,
                   int32_t* chunks_out) {
    if (!h || (qb > 0 && (!queries || !best_class))) return fpnn_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return fpnn_fail(FIR_ERR_ARG, "qb < 0");
    FPNN_HIP(hipSetDevice(h->device));
    const int nsub = (int)std::max<size_t>(1, std::min<size_t>(kChunk, (60 * 1024 - (size_t)h->C * 5) / ((size_t)h->C * 4)));
    const size_t lds = (size_t)h->C * 5 + (size_t)nsub * h->C * 4;
    double* pq = (double*)h->pin;
    float* pouts = (float*)(pq + (size_t)kPredBatch * h->d);
    int32_t* pbest = (int32_t*)(pouts + (size_t)kPredBatch * h->C);
    int32_t* pchunks = pbest + kPredBatch;
    unsigned long long* pticket = (unsigned long long*)(((uintptr_t)(pchunks + kPredBatch) + 63) & ~(uintptr_t)63);
    for (int q0 = 0; q0 < qb; q0 += kPredBatch) {
        const int nq = std::min(kPredBatch, qb - q0);
        const bool one = qb == 1;
        const unsigned long long ticket = one ? ++h->ticket : 0;
        std::memcpy(pq, queries + (size_t)q0 * h->d, (size_t)nq * h->d * 8);
        const int F = std::min(16, std::max(1, kBlock / h->C));   // features per workgroup of the term pass: about one term per thread
        hipLaunchKernelGGL(k_fpnn_terms, dim3((unsigned)((h->d + F - 1) / F), (unsigned)nq), dim3(kBlock), (size_t)F * 2 * h->J * 8, h->stream, pq,
                           h->d, h->C, h->J, F, h->avg.as<double>(), h->sd.as<double>(), h->scale, h->at.as<double>(), h->terms.as<float>());
        FPNN_HIP(hipGetLastError());
        if (seq)
            hipLaunchKernelGGL(k_fpnn_predict<true>, dim3(nq), dim3(kBlock), lds, h->stream, h->d, h->C, h->terms.as<float>(), output_ratio, nsub, pbest, outputs ? pouts : nullptr, pchunks, one ? pticket : nullptr, ticket);
        else
            hipLaunchKernelGGL(k_fpnn_predict<false>, dim3(nq), dim3(kBlock), lds, h->stream, h->d, h->C, h->terms.as<float>(), output_ratio, nsub, pbest, outputs ? pouts : nullptr, pchunks, one ? pticket : nullptr, ticket);
        FPNN_HIP(hipGetLastError());
        if (one) {
            // tens of microseconds: spin on the pinned ticket (2 ms at most), then fall back to the stream
            volatile unsigned long long* flag = pticket;
            const auto t0 = std::chrono::steady_clock::now();
            for (int spins = 0; __atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket; ++spins) {
                if ((spins & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) {
                    FPNN_HIP(hipStreamSynchronize(h->stream));
                    if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != ticket) return fpnn_fail(FIR_ERR_HIP, "the result ticket was not published");
                    break;
                }
            }
        } else {
            FPNN_HIP(hipStreamSynchronize(h->stream));
        }
        std::memcpy(best_class + q0, pbest, (size_t)nq * 4);
        if (outputs) std::memcpy(outputs + (size_t)q0 * h->C, pouts, (size_t)nq * h->C * 4);
        if (chunks_out) std::memcpy(chunks_out + q0, pchunks, (size_t)nq * 4);
    }
    return FIR_OK;
}
}  // namespace

extern "C" {

int fir_fpnn_train(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes, const double* avg,
                   const double* sd, double scale, int32_t device, fir_fpnn** out) {
    if (!out) return fpnn_fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nt <= 0 || d <= 0 || num_classes <= 0 || !train_rows || !train_class || !avg || !sd)
        return fpnn_fail(FIR_ERR_ARG, "bad arguments (nt=%lld d=%d classes=%d)", (long long)nt, d, num_classes);
    if (nt >= ((int64_t)1 << 31) - 64) return fpnn_fail(FIR_ERR_ARG, "nt too large");
    if ((size_t)num_classes * 9 > 60 * 1024) return fpnn_fail(FIR_ERR_ARG, "num_classes=%d exceeds the LDS score table (6826)", num_classes);
    std::vector<int32_t> off((size_t)num_classes + 1, 0);
    for (int64_t t = 0; t < nt; ++t) {
        const int32_t cl = train_class[t];
        if (cl < 0 || cl >= num_classes || (t > 0 && cl < train_class[t - 1]))
            return fpnn_fail(FIR_ERR_ARG, "train_class must be non-decreasing in [0,%d) (row %lld)", num_classes, (long long)t);
        off[(size_t)cl + 1]++;
    }
    for (int i = 0; i < num_classes; ++i) off[(size_t)i + 1] += off[(size_t)i];
    // classification.cpp:669-676: J = ceil(cbrt(training rows per class)), at least 3
    int J = (int)std::ceil(std::pow(1.0 * (double)nt / num_classes, 1.0 / 3));
    if (J <= 3) J = 3;
    if (J > kMaxJ) return fpnn_fail(FIR_ERR_ARG, "J=%d harmonics exceed the supported %d", J, kMaxJ);
    int cnt = 0;
    cnt = fir_device_count();        // the guarded first touch of the runtime (fir_runtime_init_)
    if (cnt <= 0) return fpnn_fail(FIR_ERR_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= cnt) return fpnn_fail(FIR_ERR_NODEVICE, "device %d out of range (%d visible)", device, cnt);
    { const int rc0 = fir_runtime_init_(device); if (rc0) return rc0; }
    hipDeviceProp_t prop;
    FPNN_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fpnn_fail(FIR_ERR_NODEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    fir_fpnn* h = new (std::nothrow) fir_fpnn();
    if (!h) return fpnn_fail(FIR_ERR_NOMEM, "host allocation failed");
    struct Guard { fir_fpnn* h; ~Guard() { if (h) fir_fpnn_destroy(h); } } guard{h};
    h->device = device; h->d = d; h->C = num_classes; h->J = J; h->scale = scale;
    const int K = 2 * J + 1;
    FPNN_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    FPNN_HIP(h->at.alloc((size_t)d * K * num_classes * 8));
    FPNN_HIP(h->avg.alloc((size_t)d * 8));
    FPNN_HIP(h->sd.alloc((size_t)d * 8));
    FPNN_HIP(h->terms.alloc((size_t)kPredBatch * d * num_classes * 4));
    FPNN_HIP(hipHostMalloc(&h->pin, (size_t)kPredBatch * ((size_t)d * 8 + (size_t)num_classes * 4 + 8) + 128, hipHostMallocDefault));
    std::memset((char*)h->pin + (size_t)kPredBatch * ((size_t)d * 8 + (size_t)num_classes * 4 + 8), 0, 128);
    Buf drows, doff;
    FPNN_HIP(drows.alloc((size_t)nt * d * 8));
    FPNN_HIP(doff.alloc(off.size() * 4));
    FPNN_HIP(hipMemcpyAsync(drows.p, train_rows, (size_t)nt * d * 8, hipMemcpyHostToDevice, h->stream));
    FPNN_HIP(hipMemcpyAsync(doff.p, off.data(), off.size() * 4, hipMemcpyHostToDevice, h->stream));
    FPNN_HIP(hipMemcpyAsync(h->avg.p, avg, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
    FPNN_HIP(hipMemcpyAsync(h->sd.p, sd, (size_t)d * 8, hipMemcpyHostToDevice, h->stream));
    const int64_t total = (int64_t)num_classes * J * d;
    hipLaunchKernelGGL(k_fpnn_train, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, drows.as<double>(), d,
                       doff.as<int32_t>(), num_classes, J, h->avg.as<double>(), h->sd.as<double>(), scale, h->at.as<double>());
    FPNN_HIP(hipGetLastError());
    FPNN_HIP(hipStreamSynchronize(h->stream));
    guard.h = nullptr;
    *out = h;
    return FIR_OK;
}

int fir_fpnn_destroy(fir_fpnn* h) {
    if (!h) return FIR_OK;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return FIR_OK;
}

int fir_fpnn_info(const fir_fpnn* h, int32_t* J, int32_t* d, int32_t* num_classes) {
    if (!h) return fpnn_fail(FIR_ERR_ARG, "NULL handle");
    if (J) *J = h->J;
    if (d) *d = h->d;
    if (num_classes) *num_classes = h->C;
    return FIR_OK;
}

int fir_fpnn_get_model(fir_fpnn* h, double* a_out) {
    if (!h || !a_out) return fpnn_fail(FIR_ERR_ARG, "NULL argument");
    FPNN_HIP(hipSetDevice(h->device));
    const int K = 2 * h->J + 1;
    const int64_t total = (int64_t)h->d * h->C * K;
    Buf da;
    FPNN_HIP(da.alloc((size_t)total * 8));
    hipLaunchKernelGGL(k_fpnn_untranspose, dim3((unsigned)((total + kBlock - 1) / kBlock)), dim3(kBlock), 0, h->stream, h->at.as<double>(), h->d, h->C,
                       K, da.as<double>());
    FPNN_HIP(hipGetLastError());
    FPNN_HIP(hipMemcpyAsync(a_out, da.p, (size_t)total * 8, hipMemcpyDeviceToHost, h->stream));
    FPNN_HIP(hipStreamSynchronize(h->stream));
    return FIR_OK;
}

int fir_fpnn_predict(fir_fpnn* h, const double* queries, int32_t qb, int32_t* best_class, float* outputs) {
    return predict_common(h, queries, qb, false, 1.0f, best_class, outputs, nullptr);
}

int fir_fpnn_predict_seq(fir_fpnn* h, const double* queries, int32_t qb, float output_ratio, int32_t* best_class, int32_t* chunks_out) {
    return predict_common(h, queries, qb, true, output_ratio, best_class, nullptr, chunks_out);
}

}  // extern "C"
