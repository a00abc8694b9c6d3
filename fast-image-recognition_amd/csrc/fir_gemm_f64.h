// fir_gemm_f64.h -- the matrix-core nomination over a FLOAT64 training set (included at the end of fir_gemm.hip): what
// KNNClassifier::predict (qt_cpp/classification.cpp:116-170) needs of 10^6 rows is the handful of nearest ones -- the walk over
// the sorted distances stops as soon as one class has K votes (:154-160) -- so large kNN batches go the way of the L2 matcher:
//   * an fp16 fragment copy of the centred rows (g - avg, what Classifier::normalize makes of a training row, :103-105, :132)
//     and the same k_gemm_proxy_f16x passes (the threshold found on the way: <3, *> for one row, <4, *> for K' rows) nominate;
//   * k_gemm_rerank_f64 recomputes every nominated row inside the rounding window of the K'-th smallest proxy with the
//     reference's own arithmetic in double -- diff = (g - avg) - (q - avg), dist += diff * diff, feature by feature, un-fused,
//     one division by the feature count (:123-143): the bits of k_cls_scan and of the oracle -- and keeps the K' nearest as
//     (distance, row), ascending;
//   * the same certificate as the f32 path (fir_gemm.hip, "The error bound E"): every row NOT re-ranked has a proxy >= p_excl,
//     hence a reference distance >= (|q'|^2 + p_excl)/d - E, E = e_rel (|q'|^2 + max |g'|^2)/d with the same e_rel = 8 d 2^-24 +
//     2^-10 (1 + 2^-4): the operand rounding to fp16 is the same 11 bits whether the value came from a float or a double, the f32
//     norms of double rows add one more 2^-24 each (inside the 8 d 2^-24 term's slack of (4 d - 10) 2^-24), the exact side's own
//     roundings are 2^-53-sized. Certified = no row outside the re-ranked set can be among the K' nearest or tie with the K'-th.
// The caller (fir_cls.hip) turns the K' rows into votes; a query that is not certified, whose K' rows do not settle the vote, or
// that has equal distances among them takes the exact f64 scan as before.
#pragma once

// max |x| over the tiled f64 rows (non-finite -> +inf), as k_gemm_absmax
__global__ void __launch_bounds__(256) k_gemm_absmax_f64(const double2* __restrict__ gal2, int64_t count2, float* __restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < count2; i += (int64_t)gridDim.x * 256) {
        const double2 g = gal2[i];
        const bool bad = !(g.x == g.x && g.y == g.y);
        const float a = (float)fmax(fabs(g.x), fabs(g.y));              // (rounding up to the next float at most: the scale has a binade of room)
        m = fmaxf(m, bad ? __builtin_huge_valf() : a);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned int*)out, __float_as_uint(m));
}

// tiled f64 rows (fir_cls.hip layout: gal2[(tile * dp2 + c) * 64 + r] = features 2c, 2c + 1 of row 64 tile + r) * scale -> the 16-row
// fp16 fragment order of fir_gemm_f16x.h
__global__ void __launch_bounds__(256) k_gemm_pack_gallery_f16x_f64(const double2* __restrict__ gal2, int64_t n, int dp2, int dk16, float scale,
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
    for (int j = 0; j < 8; j += 2) {
        const int k = 32 * kk + 8 * (l >> 4) + j;                     // even: features k, k + 1 are one double2
        double2 g = make_double2(0.0, 0.0);
        if (row < n && k < kmax) g = gal2[((row >> 6) * dp2 + (k >> 1)) * 64 + (row & 63)];
        v[j] = (_Float16)((float)g.x * scale);
        v[j + 1] = (_Float16)(k + 1 < kmax ? (float)g.y * scale : 0.f);
    }
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    gh[o] = u;
}

// gnorm[row] = |g'|^2 as a float (summed in double: the result is within 2^-24 of the true norm)
__global__ void __launch_bounds__(256) k_gemm_row_norms_f64(const double2* __restrict__ gal2, int64_t n, int dp2, float* __restrict__ gnorm) {
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n) return;
    double s = 0.0;
    for (int c = 0; c < dp2; ++c) {
        const double2 g = gal2[((row >> 6) * dp2 + c) * 64 + (row & 63)];
        s += g.x * g.x + g.y * g.y;
    }
    gnorm[row] = (float)s;
}

// k_gemm_qprep_f16 for centred float64 queries qc[nq][d] (q - avg)
__global__ void __launch_bounds__(64) k_gemm_qprep_f16_f64(const double* __restrict__ q, int nq, int d, int gallery_exp, float* __restrict__ qnorm,
                                                            float* __restrict__ qmul, float* __restrict__ qinv, int* __restrict__ counts,
                                                            float* __restrict__ win, unsigned int* __restrict__ t_bits,
                                                            const float* __restrict__ gnorm_max_p, float e_rel, int nslot) {
    const int qi = blockIdx.x;
    double s = 0.0;
    float m = 0.f;
    bool bad = false;
    if (qi < nq)
        for (int k = threadIdx.x; k < d; k += 64) {
            const double x = q[(size_t)qi * d + k];
            s += x * x;
            m = fmaxf(m, (float)fabs(x));
            bad = bad || !(fabs(x) < (double)__builtin_huge_valf());
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
        const float qn = (bad && qi < nq) ? __builtin_nanf("") : (float)s;
        qnorm[qi] = qn;
        qmul[qi] = bad ? 0.f : ldexpf(1.0f, sh);
        qinv[qi] = qi >= nq ? 0.f : bad ? __builtin_nanf("") : ldexpf(1.0f, -sh - gallery_exp);
        counts[qi] = 0;
        const float w = 2.5f * e_rel * (qn + gnorm_max_p[0]);
        win[qi] = qi >= nq ? 0.f : w + fabsf(w) * 1e-6f + 1e-30f;
        if (t_bits) {
            const unsigned int start = qi >= nq ? 0u : 0x7F800000u;
            if (nslot > 0) { for (int sl = 0; sl < 8; ++sl) atomicExch(&t_bits[(size_t)qi * 8 + sl], start); }
            else atomicExch(&t_bits[qi], start);
        }
    }
}

// k_gemm_pack_queries_f16x for float64 queries
__global__ void __launch_bounds__(256) k_gemm_pack_queries_f16x_f64(const double* __restrict__ q, int nq, int d, int dk16, const float* __restrict__ qmul,
                                                                     uint4* __restrict__ qh) {
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
        const float x = (qi < nq && k < d) ? (float)q[(size_t)qi * d + k] : 0.f;      // (double -> float -> fp16: the fp16 rounding of the float is within 2^-11 (1 + 2^-13) of the double)
        v[j] = (_Float16)(x * mul);
    }
    uint4 u;
    __builtin_memcpy(&u, &v, 16);
    qh[o] = u;
}

// The in-flight threshold SEEDED from a row sample: classifier training sets are class-major with hundreds of rows per class, all of them
// within one rounding window of each other for a query of that class. Started from +inf, every workgroup would append the whole class of
// each successive "nearest class so far" until the query's own class has been seen somewhere (1M x 512, 1000 rows per class: 2 800 - 3 900
// appended rows per query, a third of the lists over their 4 096 entries -- profiles/r04_knn_matrix_cores.txt). A strided n / 32 sample meets
// every class of 32 rows or more: T starts at (smallest sampled proxy [K-th smallest of 64 disjoint subsets' smallest] + |q|^2) + window, a
// valid bound like any other the pass holds, and only falls from there. t_bits was preset by the query preparation (+inf, padding 0).
__global__ void __launch_bounds__(256) k_gemm_seed_T(const unsigned int* __restrict__ smin, int sub_stride, int k, unsigned int* __restrict__ t_bits,
                                                      const float* __restrict__ qnorm, const float* __restrict__ win, int nq_valid) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq_valid) return;
    unsigned int o;
    if (sub_stride == 0) o = smin[q];
    else {
        unsigned int best[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) best[i] = 0xFFFFFFFFu;
        for (int sb = 0; sb < kRtSubsets; ++sb) {
            unsigned int v = smin[(size_t)sb * sub_stride + q];
            if (v == 0xFF800000u) continue;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool sw = v < best[i];
                const unsigned int t = best[i];
                best[i] = sw ? v : t;
                v = sw ? t : v;
            }
        }
        o = 0xFF800000u;
#pragma unroll
        for (int i = 0; i < 8; ++i) o = (i == k - 1 && best[i] != 0xFFFFFFFFu) ? best[i] : o;
    }
    if (o == 0xFF800000u) return;                                        // nothing sampled: the pass starts from +inf
    const float tn = fmaxf((fir::f32_from_orderable(o) + qnorm[q]) + win[q], 0.f);
    if (!(tn == tn)) return;
    if (sub_stride == 0) atomicMin(&t_bits[q], __float_as_uint(tn));
    else for (int sl = 0; sl < 8; ++sl) atomicMin(&t_bits[(size_t)q * 8 + sl], __float_as_uint(tn));
}

// Per query (one wave): the K' nearest rows among the nominated ones in the reference's float64 arithmetic + the certificate.
// out_rows[q * kp + r] / out_dist: ascending by (distance, row), -1 / +inf where fewer than K' rows were re-ranked; ok[q] = 1 when no row
// outside the re-ranked set can be among the K' nearest or tie with the K'-th.
// Dynamic LDS: dp2 double2 -- the centred query.
constexpr int kF64Group = 4;
__global__ void __launch_bounds__(64) k_gemm_rerank_f64(const unsigned long long* __restrict__ lists, const int* __restrict__ counts, const float* __restrict__ tau,
                                                         const double2* __restrict__ gal2, const double* __restrict__ queries, const float* __restrict__ qnorm,
                                                         const float* __restrict__ gnorm_max_p, int64_t n, int d, int dp2, float e_rel, int kp, int ngroup,
                                                         int32_t* __restrict__ out_rows, double* __restrict__ out_dist, int32_t* __restrict__ ok) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int cnt = counts[q];
    const int have = cnt < kListCap ? cnt : kListCap;
    const unsigned long long* L = lists + (size_t)q * kListCap;
    // the K'-th smallest list entry (keys are unique: the row is part of the key)
    unsigned long long prev = 0, kth = kKeyNone;
    int found = 0;
    for (int r = 0; r < kp; ++r) {
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
    const float E = e_rel * (qn + gmax) / (float)d;
    const float pk = found == kp ? fir::f32_from_orderable((uint32_t)(kth >> 32)) : __builtin_huge_valf();   // a short list is re-ranked whole
    float win = pk + 2.0f * E * (float)d;
    win += fabsf(win) * 1e-6f;
    // LDS: the centred query (dp2 double2; every lane reads the same element: a broadcast). Every lane re-ranks ONE candidate, straight from the
    // tiled rows with eight 16-byte loads in flight -- candidates of a class-major training set are runs of consecutive rows of one tile, whose
    // loads coalesce -- and sums it in feature order; up to 64 candidates at a time (staging four at a time in LDS took 19 of the 22 ms of a
    // 4 096-query call whose classes of 1 000 rows lie inside one rounding window: profiles/r04_knn_matrix_cores.txt).
    extern __shared__ __attribute__((aligned(16))) double2 qrow[];
    for (int c = lane; c < dp2; c += 64) {
        const int k0 = 2 * c;
        qrow[c] = make_double2(k0 < d ? queries[(size_t)q * d + k0] : 0.0, k0 + 1 < d ? queries[(size_t)q * d + k0 + 1] : 0.0);
    }
    __shared__ double bd_s[8];
    __shared__ int br_s[8];
    if (lane < 8) { bd_s[lane] = __builtin_huge_val(); br_s[lane] = -1; }
    __syncthreads();
    float p_out = __builtin_huge_valf();            // smallest proxy NOT re-ranked
    int reranked = 0;
    (void)ngroup;
    for (int base = 0; base < have; base += 64) {
        const int i = base + lane;
        const unsigned long long v = i < have ? L[i] : kKeyNone;
        const float p = fir::f32_from_orderable((uint32_t)(v >> 32));
        const bool in = i < have && p <= win;
        if (i < have && !in) p_out = fminf(p_out, p);
        const unsigned long long mask = __ballot(in);
        if (!mask) continue;                                            // wave-uniform
        reranked += __popcll(mask);
        double dist = __builtin_huge_val();
        int row = -1;
        if (in) {
            row = (int)(uint32_t)(v & 0xFFFFFFFFull);
            const double2* gr = gal2 + (size_t)(row >> 6) * dp2 * 64 + (row & 63);
            double acc = 0.0;
            for (int c0 = 0; c0 < dp2; c0 += 8) {
                double2 g8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g8[u] = gr[(size_t)(c0 + u < dp2 ? c0 + u : dp2 - 1) * 64];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (c0 + u < dp2) {
                        const double2 q2 = qrow[c0 + u];
                        const double d0 = g8[u].x - q2.x;                // classification.cpp:132-137: (g - avg) - (q - avg)
                        acc = acc + d0 * d0;                             // :141 (un-fused: the translation unit is built -ffp-contract=off)
                        const double d1 = g8[u].y - q2.y;                // (a padding feature past d is 0 - 0: + 0.0)
                        acc = acc + d1 * d1;
                    }
                }
            }
            dist = acc / (double)d;                                      // :143
            if (!(dist == dist)) { dist = __builtin_huge_val(); row = -1; }   // (a NaN distance never enters: the reference's sort would put it anywhere)
        }
        // merge this batch into the K' best (distance, row): up to K' rounds of "the batch's smallest not yet taken"; a round whose candidate
        // does not beat the current K'-th ends the merge
        for (int r = 0; r < kp; ++r) {
            double md = dist;
            int mr = row;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double od = __shfl_xor(md, off, 64);
                const int orow = __shfl_xor(mr, off, 64);
                if (orow >= 0 && (mr < 0 || od < md || (od == md && orow < mr))) { md = od; mr = orow; }
            }
            if (mr < 0) break;                                           // (wave-uniform: md / mr are the same in every lane)
            const bool beats = br_s[kp - 1] < 0 || md < bd_s[kp - 1] || (md == bd_s[kp - 1] && mr < br_s[kp - 1]);
            if (!beats) break;
            if (lane == 0) {
                double cd = md;
                int cr = mr;
                for (int j = 0; j < kp; ++j) {                           // sorted insert; empty slots are (+inf, -1)
                    const bool sw = br_s[j] < 0 || cd < bd_s[j] || (cd == bd_s[j] && cr < br_s[j]);
                    if (sw) {
                        const double td = bd_s[j];
                        const int tr = br_s[j];
                        bd_s[j] = cd; br_s[j] = cr;
                        cd = td; cr = tr;
                        if (cr < 0) break;
                    }
                }
            }
            if (row == mr) { row = -1; dist = __builtin_huge_val(); }    // taken (rows are unique within a list)
            __syncthreads();                                             // (one wave: orders lane 0's LDS writes before the next round's reads)
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) p_out = fminf(p_out, __shfl_xor(p_out, off, 64));
    if (lane == 0) {
        for (int j = 0; j < kp; ++j) { out_rows[(size_t)q * kp + j] = br_s[j]; out_dist[(size_t)q * kp + j] = bd_s[j]; }
        bool certified = false;
        if (cnt <= kListCap) {
            const float t = tau[q];
            const float p_excl = t != t ? t : fminf(p_out, t);
            const double lower = ((double)qn + (double)p_excl) / (double)d - (double)E;
            const double bd = br_s[kp - 1] >= 0 ? bd_s[kp - 1] : __builtin_huge_val();     // fewer than K' rows: any row outside would belong to the answer
            certified = lower > bd;                                  // false for NaN
            if (n <= reranked) certified = true;                      // every row was re-ranked
        }
        ok[q] = certified ? 1 : 0;
    }
}

// ---- host side ----
// The nomination state of a float64 training set that lives elsewhere (fir_cls.hip owns gal2): fp16 fragments, row norms, one scratch
// set. A fir_gemm with f64 = true; only fir_gemm_knn_f64_ and fir_gemm_destroy take it.
extern "C" int fir_gemm_create_f64_(int device, int cus, void* stream, const void* gal2v, int64_t nt, int d, int dp2, fir_gemm** out) {
    if (!out || !gal2v || nt <= 0 || d <= 0) return gemm_fail(FIR_ERR_ARG, "bad argument");
    *out = nullptr;
    if (!(cus >= 8 && (cus & 7) == 0)) return gemm_fail(FIR_ERR_ARG, "the 16-row kernels want CUs in eights (%d)", cus);
    const size_t row_lds = (size_t)dp2 * sizeof(double2);              // the re-rank keeps the centred query in LDS
    const int ngroup = 1;
    if (row_lds > kRerankLdsMax) return gemm_fail(FIR_ERR_ARG, "rows of %d float64 features are too long for the matrix-core path's re-rank", d);
    fir_gemm* m = new (std::nothrow) fir_gemm();
    if (!m) return gemm_fail(FIR_ERR_NOMEM, "host allocation failed");
    m->f64 = true;
    m->gal2 = (const double2*)gal2v;
    m->dp2 = dp2;
    m->v.device = device; m->v.cus = cus; m->v.n = nt; m->v.d = d; m->v.metric = 0; m->v.row_offset = 0; m->v.cls = nullptr; m->v.stream = (hipStream_t)stream;
    m->feat = d;
    m->precision = FIR_GEMM_F16;
    m->mfma16 = 1;
    m->rerank_group = ngroup;
    m->dk16 = (d + 16 * kRing - 1) / (16 * kRing) * kRing;
    if (const char* w = fir_knob_("FIR_GEMM_ADAPTIVE")) m->adaptive = std::atoi(w);
    if (const char* w = fir_knob_("FIR_GEMM_ADAPTIVE_TOPK")) m->adaptive_topk = std::atoi(w) != 0;
#ifdef FIR_AUDIT      // (the audit build only: shrinks the certificate's bound -- tests/test_gpu_cls.py shows that the suite can see an unsound one)
    if (const char* w = fir_knob_("FIR_GEMM_EREL_SCALE")) m->erel_scale = (float)std::atof(w);
#endif
    hipError_t e = hipSetDevice(device);
    const int64_t rblocks = (nt + 31) / 32;
    if (e == hipSuccess) e = hipMalloc((void**)&m->gh, (size_t)rblocks * m->dk16 * 64 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void**)&m->gnorm, (size_t)nt * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->gmax, 16);
    const int b = 0;
    if (e == hipSuccess) e = hipMalloc((void**)&m->qbf[b], (size_t)kPasses * (kQT / 32) * m->dk16 * 64 * sizeof(uint4));
    if (e == hipSuccess) e = hipMalloc((void**)&m->qnorm[b], kPasses * kQT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->qmul[b], kPasses * kQT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->qinv[b], kPasses * kQT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->smin[b], (size_t)kRtSubsets * kPasses * kQT * sizeof(unsigned int));
    if (e == hipSuccess) e = hipMalloc((void**)&m->tau[b], kPasses * kQT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->awin[b], kPasses * kQT * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&m->aT[b], (size_t)8 * kPasses * kQT * sizeof(unsigned int));
    m->rt_sample_rows = (int)std::min<int64_t>(nt, std::max<int64_t>(kMinSampleRows, nt / 32));
#define FIR_X_ATTR(M, S, O) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_gemm_proxy_f16x<M, S, O>, hipFuncAttributeMaxDynamicSharedMemorySize, kHalfLds);
    FIR_X_ATTR(1, 0, 0) FIR_X_ATTR(1, 0, 1) FIR_X_ATTR(1, 1, 0) FIR_X_ATTR(1, 1, 1) FIR_X_ATTR(2, 0, 0) FIR_X_ATTR(2, 0, 1) FIR_X_ATTR(2, 1, 0) FIR_X_ATTR(2, 1, 1)
    FIR_X_ATTR(3, 0, 0) FIR_X_ATTR(3, 0, 1) FIR_X_ATTR(3, 1, 0) FIR_X_ATTR(3, 1, 1) FIR_X_ATTR(4, 0, 0) FIR_X_ATTR(4, 0, 1) FIR_X_ATTR(4, 1, 0) FIR_X_ATTR(4, 1, 1)
#undef FIR_X_ATTR
    if (e == hipSuccess && row_lds > 48 * 1024)
        e = hipFuncSetAttribute((const void*)k_gemm_rerank_f64, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRerankLdsMax);
    if (e == hipSuccess) {
        hipStream_t st = m->v.stream;
        float h_max = 0.f;
        e = hipMemsetAsync(m->gmax, 0, 16, st);
        const int64_t count2 = (int64_t)((nt + 63) / 64) * dp2 * 64;
        hipLaunchKernelGGL(k_gemm_absmax_f64, dim3((unsigned)std::min<int64_t>((count2 + 255) / 256, 4096)), dim3(256), 0, st, m->gal2, count2, m->gmax);
        if (e == hipSuccess) e = hipMemcpyAsync(&h_max, m->gmax, sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        int ex = 0;
        if (h_max > 0.f && h_max < __builtin_huge_valf()) { (void)std::frexp(h_max, &ex); m->gallery_exp = 14 - ex; }
        else m->gallery_exp = h_max > 0.f ? 1000 : 0;
        const float scale = (m->gallery_exp >= -100 && m->gallery_exp <= 100) ? std::ldexp(1.0f, m->gallery_exp) : 0.f;
        const int64_t totalh = rblocks * m->dk16 * 64;
        hipLaunchKernelGGL(k_gemm_pack_gallery_f16x_f64, dim3((unsigned)((totalh + 255) / 256)), dim3(256), 0, st, m->gal2, nt, dp2, m->dk16, scale, m->gh, d);
        hipLaunchKernelGGL(k_gemm_row_norms_f64, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, st, m->gal2, nt, dp2, m->gnorm);
        hipLaunchKernelGGL(k_gemm_max, dim3(1), dim3(256), 0, st, m->gnorm, nt, m->gmax);
        if (e == hipSuccess) e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(st);
    }
    if (e != hipSuccess) {
        const int rc = gemm_fail(e == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP, "float64 matrix-core state: %s", hipGetErrorString(e));
        m->v.stream = nullptr;
        fir_gemm_destroy(m);
        return rc;
    }
    *out = m;
    return FIR_OK;
}

// d_qc[qb][d]: the centred queries (q - avg), device; out: rows[qb][kp] (-1 padded), dist[qb][kp], ok[qb] -- device, in stream order.
// *kernel_name (may be NULL) <- the dominant kernel. The caller turns rows into class votes and sends what is not settled to the exact scan.
extern "C" int fir_gemm_knn_f64_(fir_gemm* m, const double* d_qc, int32_t qb, int32_t kp, int32_t* d_rows, double* d_dist, int32_t* d_ok, void* stream,
                                 const char** kernel_name, double* flops_per_launch, hipEvent_t* ev_pair) {
    if (!m || !m->f64 || !d_qc || !d_rows || !d_dist || !d_ok) return gemm_fail(FIR_ERR_ARG, "bad argument");
    if (kp < 1 || kp > 8 || qb < 1) return gemm_fail(FIR_ERR_ARG, "kp=%d qb=%d", kp, qb);
    GEMM_HIP(hipSetDevice(m->v.device));
    hipStream_t st = stream ? (hipStream_t)stream : m->v.stream;
    const int d = m->feat;
    const int64_t n = m->v.n;
    const int grid = m->v.cus;
    const float e_rel = m->erel_scale * (8.0f * (float)d * 5.9604645e-8f + 9.765625e-4f * 1.0625f);
    const bool streamed = m->dk16 > kSlabH;
    const bool odd = (m->dk16 / kRing) & 1;
    const int sbq = std::min(kPasses * kQT, std::max(1024, (qb + 1023) / 1024 * 1024));
    {
        const int need = (std::min(sbq, qb) + 2 * kQT - 1) / (2 * kQT) * (2 * kQT);
        if (need > m->lists_cap) {
            GEMM_HIP(hipStreamSynchronize(st));
            (void)hipFree(m->lists[0]); (void)hipFree(m->counts[0]);
            m->lists[0] = nullptr; m->counts[0] = nullptr;
            m->lists_cap = 0;
            GEMM_HIP(hipMalloc((void**)&m->lists[0], (size_t)need * kListCap * sizeof(unsigned long long)));
            GEMM_HIP(hipMalloc((void**)&m->counts[0], (size_t)need * sizeof(int)));
            m->lists_cap = need;
        }
    }
    const int b = 0;
    const int share_cap = streamed ? std::min(m->share_max, m->share_streamed) : m->share_max;
    const size_t rr_lds = (size_t)m->dp2 * sizeof(double2);
    bool first = true;
    for (int q0 = 0; q0 < qb; q0 += sbq) {
        const int nq = std::min(sbq, qb - q0);
        const int np = (nq + kQT - 1) / kQT, pairs = (np + 1) / 2;
        const double* dq = d_qc + (size_t)q0 * d;
        int Pmax = 1;
        while (Pmax * 2 <= pairs && Pmax * 2 <= share_cap) Pmax *= 2;
        // (every super-batch: the pass starts from a threshold seeded by the row sample, so it is tight from its first row block on whatever the
        // number of row groups a workgroup sees -- 128 queries per call over 1M x 512: 1.83 -> 1.45 ms; profiles/r04_adaptive_cutoff.txt)
        const bool adaptive = m->adaptive > 0 && (kp == 1 || m->adaptive_topk);
        hipLaunchKernelGGL(k_gemm_qprep_f16_f64, dim3(pairs * 2 * kQT), dim3(64), 0, st, dq, nq, d, m->gallery_exp, m->qnorm[b], m->qmul[b], m->qinv[b], m->counts[b], m->awin[b],
                           adaptive ? m->aT[b] : (unsigned int*)nullptr, (const float*)m->gmax, e_rel, kp > 1 ? kp : 0);
        hipLaunchKernelGGL(k_gemm_pack_queries_f16x_f64, dim3((4 * m->dk16 * 64 + 255) / 256, pairs), dim3(256), 0, st, dq, nq, d, m->dk16, (const float*)m->qmul[b], m->qbf[b]);
        const int sub_stride = kp > 1 ? kPasses * kQT : 0;
        {
            // the row sample: the smallest (K'-th of 64 subsets' smallest) sampled proxy + one window is the sample flow's bound and the
            // adaptive pass's start value (k_gemm_seed_T)
            GEMM_HIP(hipMemsetD32Async((hipDeviceptr_t)m->smin[b], (int)0xFF800000u, sub_stride ? (size_t)kRtSubsets * sub_stride : (size_t)pairs * 2 * kQT, st));
            const int64_t sample_blocks = ((int64_t)m->rt_sample_rows + 31) / 32;
            const int rb_stride = (int)std::max<int64_t>(1, ((n + 31) / 32) / sample_blocks);
            for (int p0 = 0; p0 < pairs;) {
                int P = 1;
                while (P * 2 <= pairs - p0 && P * 2 <= m->share_max) P *= 2;
                const size_t qo = (size_t)p0;
                hipLaunchKernelGGL(pick_x(2, streamed, odd), dim3(grid), dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT, n,
                                   (int64_t)0, sample_blocks * 32, m->dk16, m->tau[b], m->lists[b], m->counts[b], (float*)nullptr, 0, P, P <= 1 ? 1 : 0, rb_stride,
                                   m->smin[b] + qo * 2 * kQT, sub_stride);
                p0 += P;
            }
            if (adaptive)
                hipLaunchKernelGGL(k_gemm_seed_T, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, st, (const unsigned int*)m->smin[b], sub_stride, kp, m->aT[b], (const float*)m->qnorm[b],
                                   (const float*)m->awin[b], nq);
            else if (sub_stride)
                hipLaunchKernelGGL(k_gemm_tau_kmin, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, st, m->smin[b], sub_stride, kp, m->tau[b], pairs * 2 * kQT, nq, m->qnorm[b], m->gmax, e_rel);
            else
                hipLaunchKernelGGL(k_gemm_tau_min, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, st, m->smin[b], m->tau[b], pairs * 2 * kQT, nq, m->qnorm[b], m->gmax, e_rel);
        }
        for (int p0 = 0; p0 < pairs;) {
            int P = 1;
            while (P * 2 <= pairs - p0 && P * 2 <= share_cap) P *= 2;
            const size_t qo = (size_t)p0;
            const int nt_flag = P <= 1 ? 1 : 0;
            if (ev_pair && first) GEMM_HIP(hipEventRecord(ev_pair[0], st));
            if (adaptive)
                hipLaunchKernelGGL(pick_x(kp > 1 ? 4 : 3, streamed, odd), dim3(grid, 1), dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT,
                                   n, (int64_t)0, n, m->dk16, m->awin[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT, m->qnorm[b] + qo * 2 * kQT, 0,
                                   P, nt_flag, 1, m->aT[b] + qo * 2 * kQT * (kp > 1 ? 8 : 1), kp > 1 ? kp : 0);
            else
                hipLaunchKernelGGL(pick_x(1, streamed, odd), dim3(grid, 1), dim3(kGemmBlock), kHalfLds, st, m->gh, m->gnorm, m->qbf[b] + qo * 4 * m->dk16 * 64, m->qinv[b] + qo * 2 * kQT, n,
                                   (int64_t)0, n, m->dk16, m->tau[b] + qo * 2 * kQT, m->lists[b] + qo * 2 * kQT * kListCap, m->counts[b] + qo * 2 * kQT, (float*)nullptr, 0, P, nt_flag, 1,
                                   (unsigned int*)nullptr, 0);
            if (ev_pair && first) {
                GEMM_HIP(hipEventRecord(ev_pair[1], st));
                if (flops_per_launch) *flops_per_launch = 2.0 * (double)n * d * 128.0 * P;
                if (kernel_name) *kernel_name = name_x(streamed, odd, adaptive, kp > 1);
                first = false;
            }
            p0 += P;
        }
        if (adaptive)
            hipLaunchKernelGGL(k_gemm_adapt_final, dim3((pairs * 2 * kQT + 255) / 256), dim3(256), 0, st, m->aT[b], m->qnorm[b], m->tau[b], pairs * 2 * kQT, nq, kp > 1 ? kp : 0);
        hipLaunchKernelGGL(k_gemm_rerank_f64, dim3(nq), dim3(64), rr_lds, st, m->lists[b], m->counts[b], m->tau[b], m->gal2, dq, m->qnorm[b], m->gmax, n, d, m->dp2, e_rel, kp,
                           m->rerank_group, d_rows + (size_t)q0 * kp, d_dist + (size_t)q0 * kp, d_ok + q0);
        m->passes += np;
#ifdef FIR_AUDIT
        if (fir_knob_("FIR_GEMM_DEBUG_COUNTS")) {      // audit builds: appended rows / certified queries of this super-batch (synchronises)
            GEMM_HIP(hipStreamSynchronize(st));
            std::vector<int> hc((size_t)nq), hok((size_t)nq);
            std::vector<float> ht((size_t)nq);
            GEMM_HIP(hipMemcpy(hc.data(), m->counts[b], (size_t)nq * sizeof(int), hipMemcpyDeviceToHost));
            GEMM_HIP(hipMemcpy(hok.data(), d_ok + q0, (size_t)nq * sizeof(int), hipMemcpyDeviceToHost));
            GEMM_HIP(hipMemcpy(ht.data(), m->tau[b], (size_t)nq * sizeof(float), hipMemcpyDeviceToHost));
            long long sum = 0; int mx = 0, over = 0, bad = 0;
            for (int i = 0; i < nq; ++i) { sum += hc[i]; mx = std::max(mx, hc[i]); over += hc[i] > kListCap; bad += hok[i] ? 0 : 1; }
            std::fprintf(stderr, "fir_gemm f64: %d queries (adaptive %d, K' %d): appended mean %.1f max %d, %d lists over %d, %d uncertified; tau[0..3] %g %g %g %g\n", nq, (int)adaptive, kp,
                         (double)sum / nq, mx, over, kListCap, bad, ht[0], ht[1 % nq], ht[2 % nq], ht[3 % nq]);
        }
#endif
    }
    GEMM_HIP(hipGetLastError());
    return FIR_OK;
}
