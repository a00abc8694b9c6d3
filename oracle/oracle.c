/*
 * oracle.c -- CPU restatement of the reference's gallery-match hot path, in plain C.
 *
 * TEST INFRASTRUCTURE ONLY. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker. The product (libfir_amd.so and the C++
 * host shim) never includes, links, loads or calls anything in oracle/.
 *
 * Parity status: PINNED. Every function here is checked (tests/test_oracle_vs_ref.py, run in
 * the build container) against the real reference code compiled from /root/reference by
 * oracle/build_ref.sh, and against the committed fixtures in tests/golden/ that the same
 * reference build generated (tests/golden/make_golden.py). The reference ships no tests,
 * golden vectors or data files of its own.
 *
 * Arithmetic contract: the reference is built for baseline x86-64 (no FMA), so every
 * float expression below is evaluated in the reference's order with one IEEE rounding per
 * operation. Build with -ffp-contract=off and without -ffast-math (oracle/Makefile).
 *
 * All reference citations are path:line under /root/reference/.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_L2 0
#define ORC_CHI2 1
#define ORC_KL 2

/* qt_cpp/db_features.cpp:22-42  feature_distance(lhs, rhs, start_pos, end_pos).
 * L2 arm :26, chi-square arm :29-31, KL/JS arm :33-36, mean over the range :40. */
float orc_feature_distance(const float* lhs, const float* rhs, int start_pos, int end_pos, int metric) {
    float dist = 0;
    for (int i = start_pos; i < end_pos; ++i) {
        if (metric == ORC_L2) {
            dist += (lhs[i] - rhs[i]) * (lhs[i] - rhs[i]);
        } else if ((lhs[i] + rhs[i]) > 0) {
            if (metric == ORC_CHI2) {
                dist += (lhs[i] - rhs[i]) * (lhs[i] - rhs[i]) / (lhs[i] + rhs[i]);
            } else {
                if (lhs[i] > 0) dist += lhs[i] * logf(2 * lhs[i] / (lhs[i] + rhs[i]));
                if (rhs[i] > 0) dist += rhs[i] * logf(2 * rhs[i] / (lhs[i] + rhs[i]));
            }
        }
    }
    dist /= (end_pos - start_pos);
    return dist;
}

/* The per-row loop body of qt_cpp/db_features.cpp:325-328 for all rows: out[j] = distance(q, row j). */
void orc_all_distances(const float* rows, int64_t n, int d, const float* query, int start_pos, int end_pos,
                       int metric, float* out) {
    for (int64_t j = 0; j < n; ++j) out[j] = orc_feature_distance(query, rows + j * d, start_pos, end_pos, metric);
}

/* qt_cpp/db_features.cpp:319-335 recognize_image_bf (and the identical scan of
 * qt_cpp/ann.cpp:113-126 BruteForce::recognize): strict `<` running minimum from 100000 in row
 * order, -1 when nothing beats it. start_pos is 0 in the reference; it is a parameter here for
 * the sub-range scans of ImageTesting.cpp:174,243. */
int64_t orc_recognize_bf(const float* rows, int64_t n, int d, const float* query, int start_pos, int end_pos,
                         int metric, float* best_dist_out) {
    int64_t bestInd = -1;
    double bestDist = 100000;
    float bestF = 100000.0f;
    for (int64_t j = 0; j < n; ++j) {
        float dj = orc_feature_distance(query, rows + j * d, start_pos, end_pos, metric);
        double dist = dj; /* vector<double> distances, db_features.cpp:324 */
        if (dist < bestDist) {
            bestDist = dist;
            bestF = dj;
            bestInd = j;
        }
    }
    if (best_dist_out) *best_dist_out = bestF;
    return bestInd;
}

/* The "CPU-all" row of BASELINE.md section 3: recognize_image_bf for a batch of queries, one query per OpenMP thread at a
 * time (queries are independent; every query's arithmetic is orc_recognize_bf's, so the results are bit-identical to
 * the single-threaded reference). Returns the number of threads used. */
#ifdef _OPENMP
#include <omp.h>
#endif
int orc_recognize_bf_batch_omp(const float* rows, int64_t n, int d, const float* queries, int nq, int start_pos, int end_pos,
                               int metric, int want_threads, int64_t* idx_out, float* dist_out) {
    int threads = 1;
#ifdef _OPENMP
    threads = want_threads > 0 ? want_threads : omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
#endif
    for (int q = 0; q < nq; ++q) {
        float bd = 100000.0f;
        idx_out[q] = orc_recognize_bf(rows, n, d, queries + (int64_t)q * d, start_pos, end_pos, metric, &bd);
        dist_out[q] = bd;
    }
    return threads;
}

/* K smallest rows of the reference's distance vector (db_features.cpp:325-333 generalised from
 * 1 to K winners): ascending distance, equal distances in ascending row order, only rows with
 * distance < 100000, unused slots idx = -1 / dist = 100000. K = 1 equals orc_recognize_bf. */
void orc_topk(const float* rows, int64_t n, int d, const float* query, int start_pos, int end_pos, int metric,
              int k, int64_t* idx_out, float* dist_out) {
    for (int i = 0; i < k; ++i) { idx_out[i] = -1; dist_out[i] = 100000.0f; }
    for (int64_t j = 0; j < n; ++j) {
        float dj = orc_feature_distance(query, rows + j * d, start_pos, end_pos, metric);
        if (!(dj < dist_out[k - 1])) continue; /* strict: a later equal row never displaces an earlier one */
        int p = k - 1;
        while (p > 0 && dj < dist_out[p - 1]) { dist_out[p] = dist_out[p - 1]; idx_out[p] = idx_out[p - 1]; --p; }
        dist_out[p] = dj;
        idx_out[p] = j;
    }
}

/* qt_cpp/ImageTesting.cpp:58-71 BruteForceClassifier::recognize -> classNo of the best row or -1. */
int orc_bf_classifier(const float* rows, int64_t n, int d, const int32_t* class_no, const float* query,
                      int max_features, int metric) {
    int64_t b = orc_recognize_bf(rows, n, d, query, 0, max_features, metric, 0);
    return b == -1 ? -1 : class_no[b];
}

static int cmp_desc_double(const void* a, const void* b) {
    double x = *(const double*)a, y = *(const double*)b;
    return (x < y) - (x > y);
}

/* qt_cpp/ImageTesting.cpp:108-186 ConventionalTWDClassifier::recognize.
 * type 0 Posteriors (:118-122,141-149), 1 DistDiff (:157-159), 2 DistRatio (:160-162);
 * second stage :165-180 (last_feature = 256). The sum of the 5 largest class posteriors is
 * taken in descending order here; the reference sums them in the order std::nth_element
 * leaves them (:141-146), which can differ in the last bit of `sum`. */
int orc_twd_conventional(const float* rows, int64_t n, int d, const int32_t* class_no, const float* query,
                         int num_of_classes, int type, double threshold, int reduced_features_count, int metric,
                         int* unreliable_out) {
    int64_t bestInd = -1;
    double bestDist = 100000, secondBestDist = 100000;
    double* distances = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    const int DIST_WEIGHT = 100;
    double* probabs = (double*)calloc((size_t)(num_of_classes > 5 ? num_of_classes : 5), sizeof(double));
    double max_probab = 0, probab = 0;
    for (int64_t j = 0; j < n; ++j) {
        distances[j] = orc_feature_distance(query, rows + j * d, 0, reduced_features_count, metric);
        if (type == 0) {
            probab = exp(-distances[j] * DIST_WEIGHT);
            if (probab > probabs[class_no[j]]) probabs[class_no[j]] = probab;
        }
        if (distances[j] < bestDist) {
            if (bestInd != -1 && class_no[bestInd] != class_no[j]) secondBestDist = bestDist;
            bestDist = distances[j];
            bestInd = j;
            if (type == 0) max_probab = probab;
        }
    }
    int is_reliable = 0;
    if (type == 0) {
        const int MAX_PROBABS_COUNT = 5;
        qsort(probabs, (size_t)num_of_classes, sizeof(double), cmp_desc_double);
        double sum = 0;
        for (int i = 0; i < MAX_PROBABS_COUNT; ++i) sum += probabs[i];
        max_probab /= sum;
        is_reliable = max_probab > threshold;
    } else if (type == 1) {
        is_reliable = (secondBestDist - bestDist) > threshold;
    } else {
        is_reliable = (bestDist / secondBestDist) < threshold;
    }
    if (unreliable_out) *unreliable_out = !is_reliable;
    if (!is_reliable) {
        bestInd = -1;
        bestDist = 100000;
        int last_feature = 256;
        for (int64_t j = 0; j < n; ++j) {
            distances[j] = (distances[j] * reduced_features_count +
                            orc_feature_distance(query, rows + j * d, reduced_features_count, last_feature, metric) *
                                (last_feature - reduced_features_count)) / last_feature;
            if (distances[j] < bestDist) { bestDist = distances[j]; bestInd = j; }
        }
    }
    int bestClassInd = -1;
    if (bestInd != -1) bestClassInd = class_no[bestInd];
    free(distances);
    free(probabs);
    return bestClassInd;
}

/* qt_cpp/ImageTesting.cpp:207-288 ProposedTWDClassifier::recognize (CHECK_ALL_INSTANCES build,
 * :206): chunks of reduced_features_count dims up to 256 (:227,229), running per-row sums :243,
 * pruning by bestDist * (1/th) :254-266, stop when one class is left :278. */
int orc_twd_proposed(const float* rows, int64_t n, int d, const int32_t* class_no, const float* query,
                     int reduced_features_count, double th, int metric, int* unreliable_out, int* chunks_out) {
    double threshold = 1.0 / th; /* ImageTesting.cpp:191 */
    int64_t bestInd = -1;
    double* distances = (double*)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
    int* instances_to_check = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
    for (int64_t j = 0; j < n; ++j) instances_to_check[j] = 1;
    int last_feature = 256;
    int unreliable = 0, chunks = 0;
    for (int cur_features = 0; cur_features < last_feature; cur_features += reduced_features_count) {
        double bestDist = 100000;
        ++chunks;
        for (int64_t j = 0; j < n; ++j) {
            if (!instances_to_check[j]) continue;
            distances[j] += orc_feature_distance(query, rows + j * d, cur_features,
                                                 cur_features + reduced_features_count, metric);
            if (distances[j] < bestDist) { bestDist = distances[j]; bestInd = j; }
        }
        int num_of_variants = 0;
        double dist_threshold = bestDist * threshold;
        ++num_of_variants;
        int bestClass = class_no[bestInd];
        for (int64_t j = 0; j < n; ++j) {
            if (instances_to_check[j]) {
                if (distances[j] > dist_threshold) instances_to_check[j] = 0;
                else if (class_no[j] != bestClass) ++num_of_variants;
            }
        }
        if (num_of_variants == 1) break;
        if (cur_features == 0) ++unreliable;
    }
    if (unreliable_out) *unreliable_out = unreliable;
    if (chunks_out) *chunks_out = chunks;
    int bestClassInd = -1;
    if (bestInd != -1) bestClassInd = class_no[bestInd];
    free(distances);
    free(instances_to_check);
    return bestClassInd;
}

/* ------------------------------------------------------------------------------------------
 * Double-precision classifiers of qt_cpp/classification.cpp. The training rows are given in the
 * reference's scan order: class 0's training_set rows, then class 1's, ... (:121-122,195-198):
 * train_rows[nt][d], train_class[nt] non-decreasing, avg[d] = avgValues (:984).
 * ------------------------------------------------------------------------------------------ */

/* Mean squared distance of classification.cpp:123-143 (KNN) for one training row. */
static double cls_dist_sum(const double* g, const double* q, const double* avg, int d) {
    double dist = 0;
    for (int fi = 0; fi < d; ++fi) {
        double diff = g[fi] - avg[fi]; /* normalize(), :103-105 */
        double val = q[fi] - avg[fi];
        diff -= val;
        dist += diff * diff;
    }
    return dist;
}

/* classification.cpp:116-170 KNNClassifier::predict. dist_out (nullable) receives the nt mean
 * distances (:143). Equal distances are visited in scan order here (the reference's std::sort
 * :151 leaves their order unspecified). */
typedef struct { double d; int64_t i; } orc_di;
static int cmp_di(const void* a, const void* b) {
    const orc_di* x = (const orc_di*)a; const orc_di* y = (const orc_di*)b;
    if (x->d < y->d) return -1;
    if (x->d > y->d) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
int orc_knn_predict(const double* train_rows, const int32_t* train_class, int64_t nt, int d, const double* avg,
                    int num_of_classes, const double* q, int K, double* dist_out) {
    float* outputs = (float*)calloc((size_t)num_of_classes, sizeof(float));
    orc_di* di = (orc_di*)malloc(sizeof(orc_di) * (size_t)(nt > 0 ? nt : 1));
    for (int64_t t = 0; t < nt; ++t) {
        double dist = cls_dist_sum(train_rows + t * d, q, avg, d);
        dist /= d;
        di[t].d = dist; di[t].i = t;
        if (dist_out) dist_out[t] = dist;
    }
    qsort(di, (size_t)nt, sizeof(orc_di), cmp_di);
    for (int64_t i = 0; i < nt; ++i) {
        int c = train_class[di[i].i];
        ++outputs[c];
        if (outputs[c] >= K) break;
    }
    float max_output = -DBL_MAX; /* float max_output=-DBL_MAX -> -inf, :161 */
    int bestClass = -1;
    for (int i = 0; i < num_of_classes; ++i)
        if (max_output < outputs[i]) { max_output = outputs[i]; bestClass = i; }
    free(outputs);
    free(di);
    return bestClass;
}

/* classification.cpp:188-226 PNNClassifier::predict_bf. total_training_size = nt (:189 is
 * dataset.size()-test_set.size()). scores_out (nullable) receives the num_of_classes outputs. */
int orc_pnn_predict(const double* train_rows, const int32_t* train_class, int64_t nt, int d, const double* avg,
                    int num_of_classes, const double* q, double* scores_out) {
    double var = 0.00002;
    if (d > 2000) var /= 10;
    double* outputs = (double*)calloc((size_t)num_of_classes, sizeof(double));
    double den = (double)nt;
    for (int64_t t = 0; t < nt; ++t) {
        double dist = cls_dist_sum(train_rows + t * d, q, avg, d);
        outputs[train_class[t]] += exp(-dist / (2 * (size_t)d * var));
    }
    for (int i = 0; i < num_of_classes; ++i) outputs[i] /= den;
    double max_output = -DBL_MAX;
    int bestClass = -1;
    for (int i = 0; i < num_of_classes; ++i)
        if (max_output < outputs[i]) { max_output = outputs[i]; bestClass = i; }
    if (scores_out) memcpy(scores_out, outputs, sizeof(double) * (size_t)num_of_classes);
    free(outputs);
    return bestClass;
}

/* classification.cpp:320-388 PNNwithClusteringClassifier::train for ONE class: rows[n][d] are the class's
 * training rows (raw dataset features, no mean subtraction, :339-342) in training_set order. 100 rounds of
 * assign-to-nearest-medoid (:331-350) / re-pick the member with the smallest summed distance (:351-375).
 * medoids_out[num_clusters] <- positions inside the class (or -1 for an emptied cluster); returns how many
 * were written. Classes with n <= num_clusters keep every row (:381-385): medoids 0..n-1. */
int orc_pnn_cluster_class(const double* rows, int n, int d, int num_clusters, int* medoids_out) {
    if (n <= num_clusters) {
        for (int j = 0; j < n; ++j) medoids_out[j] = j;
        return n;
    }
    long* centroid = (long*)malloc(sizeof(long) * (size_t)num_clusters);
    long* assign = (long*)malloc(sizeof(long) * (size_t)n);
    for (int c = 0; c < num_clusters; ++c) centroid[c] = c;
    for (int step = 0; step < 100; ++step) {
        for (int t = 0; t < n; ++t) {
            assign[t] = -1;
            double bestDist = DBL_MAX;
            for (int c = 0; c < num_clusters; ++c) {
                if (centroid[c] < 0) continue; /* the reference's size_t compare is always true; an emptied cluster is UB there */
                double dist = 0;
                const double* a = rows + centroid[c] * d;
                const double* b = rows + (long)t * d;
                for (int f = 0; f < d; ++f) dist += (a[f] - b[f]) * (a[f] - b[f]);
                dist /= d;
                if (dist < bestDist) { bestDist = dist; assign[t] = c; }
            }
        }
        for (int c = 0; c < num_clusters; ++c) {
            double bestClustDist = DBL_MAX;
            centroid[c] = -1;
            for (int t = 0; t < n; ++t) {
                if (assign[t] != c) continue;
                double clustDist = 0;
                for (int t1 = 0; t1 < n; ++t1) {
                    if (assign[t1] != c) continue;
                    double dist = 0;
                    const double* a = rows + (long)t * d;
                    const double* b = rows + (long)t1 * d;
                    for (int f = 0; f < d; ++f) dist += (a[f] - b[f]) * (a[f] - b[f]);
                    dist /= d;
                    clustDist += dist;
                }
                if (clustDist < bestClustDist) { bestClustDist = clustDist; centroid[c] = t; }
            }
        }
    }
    int m = 0;
    for (int c = 0; c < num_clusters; ++c)
        if (centroid[c] >= 0) medoids_out[m++] = (int)centroid[c];
    free(centroid);
    free(assign);
    return m;
}

/* classification.cpp:389-428 PNNwithClusteringClassifier::predict: predict_bf over the medoid rows, but the
 * denominator stays the FULL training size (:390,393). */
int orc_pnn_predict_den(const double* train_rows, const int32_t* train_class, int64_t nt, int d, const double* avg,
                        int num_of_classes, const double* q, double total_training_size, double* scores_out) {
    double var = 0.00002;
    if (d > 2000) var /= 10;
    double* outputs = (double*)calloc((size_t)num_of_classes, sizeof(double));
    for (int64_t t = 0; t < nt; ++t) {
        double dist = 0;
        const double* g = train_rows + t * d;
        for (int fi = 0; fi < d; ++fi) {
            double diff = g[fi] - avg[fi];
            double val = q[fi] - avg[fi];
            diff -= val;
            dist += diff * diff;
        }
        outputs[train_class[t]] += exp(-dist / (2 * (size_t)d * var));
    }
    for (int i = 0; i < num_of_classes; ++i) outputs[i] /= total_training_size;
    double max_output = -DBL_MAX;
    int bestClass = -1;
    for (int i = 0; i < num_of_classes; ++i)
        if (max_output < outputs[i]) { max_output = outputs[i]; bestClass = i; }
    if (scores_out) memcpy(scores_out, outputs, sizeof(double) * (size_t)num_of_classes);
    free(outputs);
    return bestClass;
}

/* classification.cpp:228-295 PNNClassifier::predict_sequentional: 32-feature chunks (:182),
 * per-row running sums, class outputs with 2*var*max_fi (:266), classes below
 * max/1e9 dropped (:280-289, float threshold), stop when one class is left. */
int orc_pnn_predict_seq(const double* train_rows, const int32_t* train_class, int64_t nt, int d, const double* avg,
                        int num_of_classes, const double* q, int* chunks_out) {
    const int delta_features_count = 32;
    const int output_dividor = 1000000000;
    double var = 0.00002;
    if (d > 2000) var /= 10;
    int bestClass = -1;
    int* classes_to_check = (int*)malloc(sizeof(int) * (size_t)num_of_classes);
    for (int i = 0; i < num_of_classes; ++i) classes_to_check[i] = 1;
    double* distances = (double*)calloc((size_t)(nt > 0 ? nt : 1), sizeof(double));
    double* outputs = (double*)calloc((size_t)num_of_classes, sizeof(double));
    double den = (double)nt;
    int chunks = 0;
    for (int cur_features = 0; cur_features < d; cur_features += delta_features_count) {
        int max_fi = cur_features + delta_features_count;
        if (max_fi > d) max_fi = d;
        ++chunks;
        for (int i = 0; i < num_of_classes; ++i) if (classes_to_check[i]) outputs[i] = 0;
        for (int64_t t = 0; t < nt; ++t) {
            int c = train_class[t];
            if (!classes_to_check[c]) continue;
            const double* g = train_rows + t * d;
            for (int fi = cur_features; fi < max_fi; ++fi) {
                double diff = g[fi] - avg[fi];
                double val = q[fi] - avg[fi];
                diff -= val;
                distances[t] += diff * diff;
            }
            outputs[c] += exp(-distances[t] / (2 * var * max_fi));
        }
        for (int i = 0; i < num_of_classes; ++i) if (classes_to_check[i]) outputs[i] = outputs[i] / den;
        double max_output = -DBL_MAX;
        for (int i = 0; i < num_of_classes; ++i)
            if (classes_to_check[i] && max_output < outputs[i]) { max_output = outputs[i]; bestClass = i; }
        int num_of_variants = 0;
        float output_threshold = max_output / output_dividor;
        for (int i = 0; i < num_of_classes; ++i) {
            if (classes_to_check[i]) {
                if (outputs[i] < output_threshold) classes_to_check[i] = 0;
                else ++num_of_variants;
            }
        }
        if (num_of_variants == 1) break;
    }
    if (chunks_out) *chunks_out = chunks;
    free(classes_to_check); free(distances); free(outputs);
    return bestClass;
}

/* ---- FPNNClassifier, classification.cpp:618-791 ---- */
/* fasterlog2, classification.cpp:64-73 */
static float orc_fasterlog2(float x) {
    union { float f; uint32_t i; } vx = { x };
    union { uint32_t i; float f; } mx = { (vx.i & 0x007FFFFF) | (0x7e << 23) };
    float y = vx.i;
    y *= 1.0 / (1 << 23);
    return y - 124.22544637f - 1.498030302f * mx.f - 1.72587999f / (0.3520887068f + mx.f);
}
float orc_fastlog(float x) { return orc_fasterlog2(x); }
/* FPNNClassifier::normalize, :637-659 (the `#elif 1` arm) */
static double orc_fpnn_normalize(double x, double avg, double sd, double features_scale) {
    double val = (sd != 0) ? features_scale * (x - avg) / sd : 0;
    const double max_val = 0.5;
    if (val < -max_val) val = -max_val;
    else if (val > max_val) val = max_val;
    return val;
}
/* :669-676 */
int orc_fpnn_J(int64_t num_of_training_data, int num_of_classes) {
    int J = (int)ceil(pow(1.0 * num_of_training_data / num_of_classes, 1.0 / 3));
    const int min_J = 3;
    if (J <= min_J) J = min_J;
    return J;
}
/* FPNNClassifier::train, :661-696. train_rows class-major ([nt][d], train_class non-decreasing) = tmp_dataset rows in
 * training_set order. a[(fi*C + i)*(2J+1) + ...]. */
void orc_fpnn_train(const double* train_rows, const int32_t* train_class, int64_t nt, int d, int num_of_classes, const double* avg,
                    const double* sd, double features_scale, int J, double* a) {
    const double PI = atan(1.0) * 4;
    int64_t* start = (int64_t*)calloc((size_t)num_of_classes + 1, sizeof(int64_t));
    for (int64_t t = 0; t < nt; ++t) start[train_class[t] + 1]++;
    for (int i = 0; i < num_of_classes; ++i) start[i + 1] += start[i];
    for (int64_t k = 0; k < (int64_t)num_of_classes * d * (2 * J + 1); ++k) a[k] = 0;
    for (int fi = 0; fi < d; ++fi)
        for (int i = 0; i < num_of_classes; ++i) {
            const size_t model_ind = ((size_t)fi * num_of_classes + i) * (2 * J + 1);
            a[model_ind] = 0.5;
            const size_t sz = (size_t)(start[i + 1] - start[i]);
            const double cur_mult = 1.0 / sz;
            for (size_t t = 0; t < sz; ++t) {
                const double val = orc_fpnn_normalize(train_rows[(start[i] + (int64_t)t) * d + fi], avg[fi], sd[fi], features_scale);
                for (size_t j = 0; j < (size_t)J; ++j) {
                    a[model_ind + 2 * j + 1] += cos(PI * (j + 1) * val) * cur_mult * ((size_t)J - j) / ((size_t)J * ((size_t)J + 1));
                    a[model_ind + 2 * j + 2] += sin(PI * (j + 1) * val) * cur_mult * ((size_t)J - j) / ((size_t)J * ((size_t)J + 1));
                }
            }
        }
    free(start);
}
/* predict_bf :698-735 (seq = 0) and predict_sequentional :736-791 (seq = 1). outputs_out[C] and chunks_out nullable. */
int orc_fpnn_predict(const double* a, int J, int d, int num_of_classes, const double* avg, const double* sd, double features_scale,
                     const double* q, int seq, float output_ratio, float* outputs_out, int* chunks_out) {
    const double PI = atan(1.0) * 4;
    const float output_delta = orc_fasterlog2(output_ratio);
    const int delta_features_count = 32;
    float* outputs = (float*)calloc((size_t)num_of_classes, sizeof(float));
    int* classes_to_check = (int*)malloc(sizeof(int) * (size_t)num_of_classes);
    double* cos_vals = (double*)malloc(sizeof(double) * (size_t)J);
    double* sin_vals = (double*)malloc(sizeof(double) * (size_t)J);
    for (int i = 0; i < num_of_classes; ++i) classes_to_check[i] = 1;
    int bestClass = -1, chunks = 0;
    for (int cur_features = 0; cur_features < d; cur_features += seq ? delta_features_count : d) {
        int max_fi = seq ? cur_features + delta_features_count : d;
        if (max_fi > d) max_fi = d;
        ++chunks;
        for (int fi = cur_features; fi < max_fi; ++fi) {
            const double val = orc_fpnn_normalize(q[fi], avg[fi], sd[fi], features_scale);
            cos_vals[0] = cos(PI * val);
            sin_vals[0] = sin(PI * val);
            for (int j = 1; j < J; ++j) {
                cos_vals[j] = cos_vals[j - 1] * cos_vals[0] - sin_vals[j - 1] * sin_vals[0];
                sin_vals[j] = cos_vals[j - 1] * sin_vals[0] + sin_vals[j - 1] * cos_vals[0];
            }
            for (int i = 0; i < num_of_classes; ++i) {
                if (!classes_to_check[i]) continue;
                const size_t model_ind = ((size_t)fi * num_of_classes + i) * (2 * J + 1);
                double probab = a[model_ind];
                for (int j = 0; j < J; ++j) probab += (a[model_ind + 2 * j + 1] * cos_vals[j] + a[model_ind + 2 * j + 2] * sin_vals[j]);
                outputs[i] += orc_fasterlog2(probab);
            }
        }
        float max_output = -FLT_MAX;
        for (int i = 0; i < num_of_classes; ++i)
            if (classes_to_check[i] && max_output < outputs[i]) { max_output = outputs[i]; bestClass = i; }
        if (!seq) break;
        int num_of_variants = 0;
        const float output_threshold = max_output + output_delta * (size_t)max_fi;
        for (int i = 0; i < num_of_classes; ++i) {
            if (outputs[i] < output_threshold) classes_to_check[i] = 0;
            else ++num_of_variants;
        }
        if (num_of_variants == 1) break;
    }
    if (outputs_out) memcpy(outputs_out, outputs, sizeof(float) * (size_t)num_of_classes);
    if (chunks_out) *chunks_out = chunks;
    free(outputs); free(classes_to_check); free(cos_vals); free(sin_vals);
    return bestClass;
}

/* classification.cpp:969-989: per-feature min / max / mean / std over the training rows
 * (std = sqrt((sum x^2 - mean^2 * count) / (count - 1))). */
void orc_train_stats(const double* train_rows, int64_t nt, int d, double* mn, double* mx, double* avg, double* sd) {
    for (int fi = 0; fi < d; ++fi) {
        mn[fi] = FLT_MAX; mx[fi] = -FLT_MAX; avg[fi] = sd[fi] = 0;
        int count = 0;
        for (int64_t t = 0; t < nt; ++t) {
            ++count;
            double feature = train_rows[t * d + fi];
            if (feature < mn[fi]) mn[fi] = feature;
            if (mx[fi] < feature) mx[fi] = feature;
            avg[fi] += feature;
            sd[fi] += feature * feature;
        }
        avg[fi] /= count;
        sd[fi] = sqrt((sd[fi] - avg[fi] * avg[fi] * count) / (count - 1));
    }
}

/* ------------------------------------------------------------------------------------------
 * Feature-file format (producer qt_cpp/dnn_feature_extractor.py:58-64; consumers
 * qt_cpp/db_features.cpp:44-116 and qt_cpp/classification.cpp:795-862).
 * ------------------------------------------------------------------------------------------ */

/* `istream >> value` (libstdc++): once the stream has failed nothing is extracted; at the end of the line nothing is
 * extracted and the value is left as it was; a field that is not a number stores 0 and fails the stream. The field is
 * sign, digits, one decimal point, one exponent -- no nan / inf / hex. */
static int orc_numfield(const char* p, char* buf, size_t cap) {
    size_t len = 0; int digits = 0, point = 0;
    if (*p == '+' || *p == '-') buf[len++] = *p;
    while (len < cap - 2) {
        const char c = p[len];
        if (c >= '0' && c <= '9') digits = 1;
        else if (c == '.' && !point) point = 1;
        else break;
        buf[len++] = c;
    }
    if (!digits) return 0;
    if (p[len] == 'e' || p[len] == 'E') {
        size_t l2 = len; int ed = 0;
        buf[l2] = p[l2]; ++l2;
        if (p[l2] == '+' || p[l2] == '-') { buf[l2] = p[l2]; ++l2; }
        while (l2 < cap - 1 && p[l2] >= '0' && p[l2] <= '9') { buf[l2] = p[l2]; ++l2; ed = 1; }
        if (!ed) return 0;
        len = l2;
    }
    buf[len] = 0;
    return (int)len;
}
static void orc_extract_f32(char** pp, int* failed, float* value) {
    char buf[128];
    if (*failed) return;
    char* p = *pp + strspn(*pp, " \t\n\r\f\v");
    if (!*p) { *failed = 1; return; }
    const int len = orc_numfield(p, buf, sizeof buf);
    if (!len) { *failed = 1; *value = 0; return; }
    *value = strtof(buf, 0);
    *pp = p + len;
}
static void orc_extract_f64(char** pp, int* failed, double* value) {
    char buf[128];
    if (*failed) return;
    char* p = *pp + strspn(*pp, " \t\n\r\f\v");
    if (!*p) { *failed = 1; return; }
    const int len = orc_numfield(p, buf, sizeof buf);
    if (!len) { *failed = 1; *value = 0; return; }
    *value = strtod(buf, 0);
    *pp = p + len;
}

static char* orc_getline(FILE* f, char** buf, size_t* cap) {
    size_t len = 0;
    int c;
    while ((c = fgetc(f)) != EOF) {
        if (len + 2 > *cap) { *cap = *cap ? *cap * 2 : 1 << 16; *buf = (char*)realloc(*buf, *cap); }
        if (c == '\n') { (*buf)[len] = 0; return *buf; }
        (*buf)[len++] = (char)c;
    }
    if (len == 0) return 0;
    (*buf)[len] = 0;
    return *buf;
}

/* db_features.cpp:44-116 loadImages with d in place of FEATURES_COUNT. Rows come out in
 * ImagesDatabase order (class by first appearance :65-73, then file order). metric selects
 * the normalisation: L2 norm (:88,95) or plain sum (:91) for chi2/KL. Pass rows_out == NULL
 * to count. Returns the number of images (or -1 if rows_out is too small); 0 when the file
 * cannot be opened (:49,115). */
int64_t orc_load_images(const char* path, int d, int metric, float* rows_out, int32_t* class_out, int64_t cap_rows,
                        int* n_classes_out) {
    FILE* f = fopen(path, "r");
    if (n_classes_out) *n_classes_out = 0;
    if (!f) return 0;
    char *l1 = 0, *l2 = 0, *l3 = 0;
    size_t c1 = 0, c2 = 0, c3 = 0;
    char** names = 0; int n_names = 0, cap_names = 0;
    /* pass 1: class ids and per-class counts */
    int32_t* rec_class = 0; int64_t n_rec = 0, cap_rec = 0;
    long* rec_pos = 0;
    for (;;) {
        if (!orc_getline(f, &l1, &c1)) break;
        if (!orc_getline(f, &l2, &c2)) break;
        long pos = ftell(f);
        if (!orc_getline(f, &l3, &c3)) break;
        char* person = l2 + strspn(l2, " \t\n\r\f\v");
        if (strstr(person, "BACKGROUND_Google") || strstr(person, "257.clutter")) continue; /* :60-64 */
        int cls = -1;
        for (int i = 0; i < n_names; ++i) if (!strcmp(names[i], person)) { cls = i; break; }
        if (cls < 0) {
            if (n_names == cap_names) { cap_names = cap_names ? cap_names * 2 : 64; names = (char**)realloc(names, sizeof(char*) * (size_t)cap_names); }
            names[n_names] = strdup(person);
            cls = n_names++;
        }
        if (n_rec == cap_rec) {
            cap_rec = cap_rec ? cap_rec * 2 : 1024;
            rec_class = (int32_t*)realloc(rec_class, sizeof(int32_t) * (size_t)cap_rec);
            rec_pos = (long*)realloc(rec_pos, sizeof(long) * (size_t)cap_rec);
        }
        rec_class[n_rec] = cls; rec_pos[n_rec] = pos; ++n_rec;
    }
    if (n_classes_out) *n_classes_out = n_names;
    int64_t ret = n_rec;
    if (rows_out) {
        if (n_rec > cap_rows) ret = -1;
        else {
            /* class-major output offsets */
            int64_t* start = (int64_t*)calloc((size_t)n_names + 1, sizeof(int64_t));
            for (int64_t r = 0; r < n_rec; ++r) start[rec_class[r] + 1]++;
            for (int i = 0; i < n_names; ++i) start[i + 1] += start[i];
            for (int64_t r = 0; r < n_rec; ++r) {
                fseek(f, rec_pos[r], SEEK_SET);
                orc_getline(f, &l3, &c3);
                int64_t o = start[rec_class[r]]++;
                float* feat = rows_out + o * d;
                char* p = l3;
                float dfeature = 0, sum = 0;
                int failed = 0;
                for (int i = 0; i < d; ++i) {
                    orc_extract_f32(&p, &failed, &dfeature);               /* iss >> dfeature, :83 */
                    if (fabsf(dfeature) < 0.0001) dfeature = 0;           /* :85-86 (compared in double) */
                    feat[i] = dfeature;
                    if (metric == ORC_L2) sum += dfeature * dfeature;       /* :88 */
                    else sum += dfeature;                                  /* :91 */
                }
                if (metric == ORC_L2) sum = sqrtf(sum);                     /* :95-96 */
                for (int i = 0; i < d; ++i) feat[i] /= sum;                 /* :98-99 */
                if (class_out) class_out[o] = rec_class[r];
            }
            free(start);
        }
    }
    for (int i = 0; i < n_names; ++i) free(names[i]);
    free(names); free(rec_class); free(rec_pos); free(l1); free(l2); free(l3);
    fclose(f);
    return ret;
}

/* classification.cpp:795-862 load_image_dataset with d in place of FEATURES_COUNT: doubles,
 * no zero-clip, L2 normalisation in double (:829-847), rows kept in FILE order, class by first
 * appearance (:817-823). Returns the number of rows. */
int64_t orc_load_dataset_f64(const char* path, int d, double* rows_out, int32_t* labels_out, int64_t cap_rows,
                             int* n_classes_out) {
    FILE* f = fopen(path, "r");
    if (n_classes_out) *n_classes_out = 0;
    if (!f) return 0;
    char *l1 = 0, *l2 = 0, *l3 = 0;
    size_t c1 = 0, c2 = 0, c3 = 0;
    char** names = 0; int n_names = 0, cap_names = 0;
    int64_t n_rec = 0, ret = 0;
    for (;;) {
        if (!orc_getline(f, &l1, &c1)) break;
        if (!orc_getline(f, &l2, &c2)) break;
        if (!orc_getline(f, &l3, &c3)) break;
        char* person = l2 + strspn(l2, " \t\n\r\f\v");
        if (strstr(person, "BACKGROUND_Google") || strstr(person, "257.clutter")) continue;
        int cls = -1;
        for (int i = 0; i < n_names; ++i) if (!strcmp(names[i], person)) { cls = i; break; }
        if (cls < 0) {
            if (n_names == cap_names) { cap_names = cap_names ? cap_names * 2 : 64; names = (char**)realloc(names, sizeof(char*) * (size_t)cap_names); }
            names[n_names] = strdup(person);
            cls = n_names++;
        }
        if (rows_out) {
            if (n_rec >= cap_rows) { ret = -1; break; }
            double* feat = rows_out + n_rec * d;
            char* p = l3;
            double sum = 0;
            double v = 0;   /* `FEATURE_TYPE feature;` is declared inside the loop, uninitialised (:831): a short line reads an
                             * indeterminate value there; the same storage is reused in practice, which is what is modelled */
            int failed = 0;
            for (int i = 0; i < d; ++i) {
                orc_extract_f64(&p, &failed, &v);
                sum += v * v;
                feat[i] = v;
            }
            sum = sqrt(sum);
            for (int i = 0; i < d; ++i) feat[i] /= sum;
            if (labels_out) labels_out[n_rec] = cls;
        }
        ++n_rec;
    }
    if (n_classes_out) *n_classes_out = n_names;
    for (int i = 0; i < n_names; ++i) free(names[i]);
    free(names); free(l1); free(l2); free(l3);
    fclose(f);
    return ret < 0 ? ret : n_rec;
}

/* db_features.cpp:117-162 getTrainingAndTestImages. perm[400] is the content of `indices`
 * after the (optional) shuffle (:119-124) -- identity for randomize=false. caltech_rule != 0
 * takes 30 gallery images per class (:132-133); otherwise ceil(count*fraction) clamped to
 * [1, count-1] (:135-142). Outputs indexInDatabase (:146,150,159) and classNo per view. */
int64_t orc_split(const int32_t* class_counts, int n_classes, const int32_t* perm, int caltech_rule, double fraction,
                  int32_t* db_index, int32_t* db_class, int32_t* test_index, int32_t* test_class, int64_t* n_test_out) {
    const int INDICES_COUNT = 400;
    int64_t ndb = 0, ntest = 0;
    int indexInDatabase = 0;
    for (int class_ind = 0; class_ind < n_classes; ++class_ind) {
        int currentFaceCount = class_counts[class_ind];
        int db_size;
        if (caltech_rule) db_size = 30;
        else {
            float size_f = currentFaceCount * fraction;
            db_size = (int)(ceil(size_f));
            if (db_size == currentFaceCount) db_size = currentFaceCount - 1;
            if (db_size == 0) db_size = 1;
        }
        int ind = 0;
        for (int i = 0; i < INDICES_COUNT; ++i) {
            if (perm[i] < currentFaceCount) {
                if (ind < db_size) { db_index[ndb] = indexInDatabase + perm[i]; db_class[ndb] = class_ind; ++ndb; }
                else { test_index[ntest] = indexInDatabase + perm[i]; test_class[ntest] = class_ind; ++ntest; }
                ++ind;
            }
        }
        indexInDatabase += currentFaceCount;
    }
    if (n_test_out) *n_test_out = ntest;
    return ndb;
}

/* qt_cpp/ann.cpp:302-331 (PIVOT build of DirectedEnumeration's constructor): the pivot x gallery distance table and
 * the greedy farthest-point choice of the next pivot. pivots[0] is given (the reference draws it with random_shuffle,
 * :366-376); table[ii][j] = distance(dbImages[j], pivot ii) over all d features (ann.h:33-38: lhs = row j, rhs = the
 * pivot); min_other[ii] = smallest distance to a row of another class (:309-311, pushed to otherClassesDists :325);
 * pivots[ii+1] = the first row whose sum of distances to the pivots so far (-1000000 restarts at a pivot, :313-318)
 * is largest and > 0 (:319-322), or -1. */
void orc_dem_pivot_table(const float* rows, int64_t n, int d, const int32_t* class_no, int metric, int n_pivots, int32_t* pivots,
                         float* table, float* min_other) {
    for (int ii = 0; ii < n_pivots; ++ii) {
        const int i = pivots[ii];
        int mostFarModel = -1;
        double maxFarDist = 0;
        float min_other_dist = FLT_MAX;
        for (int64_t j = 0; j < n; ++j) {
            const float dist = orc_feature_distance(rows + j * d, rows + (int64_t)i * d, 0, d, metric);
            table[(int64_t)ii * n + j] = dist;
            if (class_no[i] != class_no[j] && dist < min_other_dist) min_other_dist = dist;
            double currentFarDist = 0;
            for (int ind = 0; ind <= ii; ++ind) {
                if (pivots[ind] == j) currentFarDist = -1000000;
                else currentFarDist += table[(int64_t)ind * n + j];
            }
            if (currentFarDist > maxFarDist) { maxFarDist = currentFarDist; mostFarModel = (int)j; }
        }
        min_other[ii] = min_other_dist;
        if (ii < n_pivots - 1) pivots[ii + 1] = mostFarModel;
    }
}

/* qt_cpp/ann.cpp:411-507 DirectedEnumeration::recognize (PIVOT build): walk the `used` kept pivots (early exit when a
 * distance drops below `threshold`, :389-399), accumulating likelihoods[nu] += (tmpDist - table[i][nu])^2 in float
 * over the index positions behind the front (:427-447, the two-write "swap" of :431-432 replayed literally); order the
 * rest by likelihood up to position image_count_to_check (:455-456) and check candidates in that order until
 * distanceCalcCount reaches image_count_to_check (:458-462). Equal likelihoods: std::partial_sort leaves their order
 * unspecified; here they keep their array order (callers avoid ties). Returns the row index or -1. */
typedef struct { float lik; int pos; int row; } orc_dem_cand;
static int orc_dem_cmp(const void* a, const void* b) {
    const orc_dem_cand* x = (const orc_dem_cand*)a; const orc_dem_cand* y = (const orc_dem_cand*)b;
    if (x->lik < y->lik) return -1;
    if (y->lik < x->lik) return 1;
    return x->pos - y->pos;
}
int orc_dem_recognize(const float* rows, int64_t n, int d, int metric, const int32_t* pivots, int used, const float* table,
                      float threshold, int image_count_to_check, const float* query, float* best_dist, int* found,
                      int* calc_count, float* lik_out) {
    float* likelihoods = (float*)calloc((size_t)n, sizeof(float));
    int* likelihood_indices = (int*)malloc((size_t)n * sizeof(int));
    int bestIndex = -1, start_index = 0, distanceCalcCount = 0, isFound = 0;
    float bestDistance = FLT_MAX, tmpDist;
    if (image_count_to_check <= 0 || image_count_to_check >= n) image_count_to_check = (int)n;   /* ann.h:20-22 */
    for (int64_t i = 0; i < n; ++i) likelihood_indices[i] = (int)i;
    for (int i = 0; i < used; ++i) {
        const int imageNum = pivots[i];
        tmpDist = orc_feature_distance(query, rows + (int64_t)imageNum * d, 0, d, metric); ++distanceCalcCount;
        if (tmpDist < bestDistance) { bestDistance = tmpDist; bestIndex = imageNum; if (bestDistance < threshold) { isFound = 1; goto end; } }
        likelihood_indices[imageNum] = likelihood_indices[start_index];
        likelihood_indices[start_index++] = imageNum;
        for (int64_t ii = start_index; ii < n; ++ii) {
            const int nu = likelihood_indices[ii];
            const float modelsDist = table[(int64_t)i * n + nu];
            if (modelsDist >= 0) { const float tmp = tmpDist - modelsDist; likelihoods[nu] += tmp * tmp; }
        }
    }
    if (lik_out) memcpy(lik_out, likelihoods, (size_t)n * sizeof(float));
    if (image_count_to_check > start_index) {
        const int64_t m = n - start_index;
        orc_dem_cand* c = (orc_dem_cand*)malloc((size_t)(m > 0 ? m : 1) * sizeof(orc_dem_cand));
        for (int64_t k = 0; k < m; ++k) { c[k].pos = (int)k; c[k].row = likelihood_indices[start_index + k]; c[k].lik = likelihoods[c[k].row]; }
        qsort(c, (size_t)m, sizeof(orc_dem_cand), orc_dem_cmp);
        for (int64_t k = 0; k < m; ++k) likelihood_indices[start_index + k] = c[k].row;
        free(c);
    }
    while (distanceCalcCount < image_count_to_check) {
        const int imageNum = likelihood_indices[start_index++];
        tmpDist = orc_feature_distance(query, rows + (int64_t)imageNum * d, 0, d, metric); ++distanceCalcCount;
        if (tmpDist < bestDistance) { bestDistance = tmpDist; bestIndex = imageNum; if (bestDistance < threshold) { isFound = 1; goto end; } }
    }
end:
    if (best_dist) *best_dist = bestDistance;
    if (found) *found = isFound;
    if (calc_count) *calc_count = distanceCalcCount;
    free(likelihoods); free(likelihood_indices);
    return bestIndex;
}

/* qt_cpp/video.cpp:35-96 loadVideos: per person a name line (:42-44), `ifs >> videos_count` (:46), per video
 * `ifs >> frames_count` and the rest of that line (:55-60), per frame a file-name line and a feature line (:63-66);
 * |x| < 1e-4 -> 0 (:74-75), sum of squares (:78), sqrt for the L2 metric only (:80-82), divide (:84-85). Persons are
 * kept sorted by name (std::map); a repeated name re-sizes and overwrites the earlier entry (:49-51).
 * Flattened like oracle/ref_wrap_match.inc::ref_load_videos_cwd. Returns the number of persons. */
typedef struct { char* name; int n_videos; int* n_frames; float*** frames; /* [video][frame] -> d floats or NULL */ } orc_person;
static void orc_skip_ws(const char** p, const char* limit) { while (*p < limit && strchr(" \t\n\r\f\v", **p)) ++*p; }
static int orc_stream_int(const char** p, const char* limit, int* fail, int* v) {
    if (*fail) return 0;
    orc_skip_ws(p, limit);
    const char* s = *p; int neg = 0; long long acc = 0;
    if (s < limit && (*s == '-' || *s == '+')) { neg = *s == '-'; ++s; }
    const char* d0 = s;
    while (s < limit && *s >= '0' && *s <= '9') acc = acc * 10 + (*s++ - '0');
    if (s == d0) { *fail = 1; *v = 0; return 0; }
    *p = s; *v = (int)(neg ? -acc : acc);
    return 1;
}
static int orc_stream_line(const char** p, const char* limit, int* fail, const char** b, const char** e) {
    if (*fail || *p >= limit) { *fail = 1; return 0; }
    *b = *p;
    const char* nl = (const char*)memchr(*p, '\n', (size_t)(limit - *p));
    *e = nl ? nl : limit;
    *p = nl ? nl + 1 : limit;
    return 1;
}
int orc_load_videos(const char* path, int d, int metric, char* names_out, int names_cap, int32_t* videos_per_person,
                    int32_t* frames_per_video, float* rows_out, int* n_videos, int* n_frames) {
    FILE* f = fopen(path, "rb");
    if (n_videos) *n_videos = 0;
    if (n_frames) *n_frames = 0;
    if (!f) return 0;
    fseek(f, 0, SEEK_END); long size = ftell(f); fseek(f, 0, SEEK_SET);
    char* text = (char*)malloc((size_t)size + 1);
    if (fread(text, 1, (size_t)size, f) != (size_t)size) { fclose(f); free(text); return 0; }
    fclose(f); text[size] = 0;
    const char* p = text; const char* limit = text + size;
    orc_person* persons = 0; int np = 0, cap = 0, fail = 0;
    while (!fail) {
        const char *b, *e;
        if (!orc_stream_line(&p, limit, &fail, &b, &e)) break;
        while (b < e && strchr(" \t\n\r\f\v", *b)) ++b;
        char* name = (char*)malloc((size_t)(e - b) + 1); memcpy(name, b, (size_t)(e - b)); name[e - b] = 0;
        int videos_count = 0;
        orc_stream_int(&p, limit, &fail, &videos_count);
        if (videos_count < 0) videos_count = 0;
        int at = 0;                                   /* sorted insert / find */
        while (at < np && strcmp(persons[at].name, name) < 0) ++at;
        if (at == np || strcmp(persons[at].name, name) != 0) {
            if (np == cap) { cap = cap ? cap * 2 : 16; persons = (orc_person*)realloc(persons, sizeof(orc_person) * (size_t)cap); }
            memmove(persons + at + 1, persons + at, sizeof(orc_person) * (size_t)(np - at));
            persons[at].name = name; persons[at].n_videos = 0; persons[at].n_frames = 0; persons[at].frames = 0; ++np;
        } else free(name);
        orc_person* P = &persons[at];
        /* resize(videos_count): existing videos beyond the new size are dropped, new ones are empty */
        P->n_frames = (int*)realloc(P->n_frames, sizeof(int) * (size_t)(videos_count > 0 ? videos_count : 1));
        P->frames = (float***)realloc(P->frames, sizeof(float**) * (size_t)(videos_count > 0 ? videos_count : 1));
        for (int i = P->n_videos; i < videos_count; ++i) { P->n_frames[i] = 0; P->frames[i] = 0; }
        P->n_videos = videos_count;
        for (int i = 0; i < videos_count; ++i) {
            int frames_count = 0;
            orc_stream_int(&p, limit, &fail, &frames_count);
            if (frames_count < 0) frames_count = 0;
            P->frames[i] = (float**)realloc(P->frames[i], sizeof(float*) * (size_t)(frames_count > 0 ? frames_count : 1));
            for (int j = 0; j < frames_count; ++j) P->frames[i][j] = 0;      /* resize(): the earlier frames are replaced */
            P->n_frames[i] = frames_count;
            if (!orc_stream_line(&p, limit, &fail, &b, &e)) break;
            for (int j = 0; j < frames_count; ++j) {
                const char *fb, *fe;
                if (!orc_stream_line(&p, limit, &fail, &b, &e)) break;
                if (!orc_stream_line(&p, limit, &fail, &fb, &fe)) break;
                char* line = (char*)malloc((size_t)(fe - fb) + 1); memcpy(line, fb, (size_t)(fe - fb)); line[fe - fb] = 0;
                char* lp = line;
                float* feat = (float*)malloc(sizeof(float) * (size_t)d);
                float dfeature = 0, sum = 0; int sfail = 0;
                for (int k = 0; k < d; ++k) {
                    orc_extract_f32(&lp, &sfail, &dfeature);
                    if (fabsf(dfeature) < 0.0001) dfeature = 0;
                    feat[k] = dfeature;
                    sum += dfeature * dfeature;
                }
                if (metric == ORC_L2) sum = sqrtf(sum);
                for (int k = 0; k < d; ++k) feat[k] /= sum;
                P->frames[i][j] = feat;
                free(line);
            }
        }
    }
    int nv = 0, nf = 0; size_t off = 0;
    for (int a = 0; a < np; ++a) {
        const size_t len = strlen(persons[a].name);
        if (names_out && off + len + 2 < (size_t)names_cap) { memcpy(names_out + off, persons[a].name, len); names_out[off + len] = '\n'; off += len + 1; names_out[off] = 0; }
        if (videos_per_person) videos_per_person[a] = persons[a].n_videos;
        for (int i = 0; i < persons[a].n_videos; ++i, ++nv) {
            if (frames_per_video) frames_per_video[nv] = persons[a].n_frames[i];
            for (int j = 0; j < persons[a].n_frames[i]; ++j, ++nf) {
                if (rows_out) {
                    if (persons[a].frames[i][j]) memcpy(rows_out + (size_t)nf * d, persons[a].frames[i][j], sizeof(float) * (size_t)d);
                    else memset(rows_out + (size_t)nf * d, 0, sizeof(float) * (size_t)d);
                }
                free(persons[a].frames[i][j]);
            }
            free(persons[a].frames[i]);
        }
        free(persons[a].frames); free(persons[a].n_frames); free(persons[a].name);
    }
    free(persons); free(text);
    if (n_videos) *n_videos = nv;
    if (n_frames) *n_frames = nf;
    return np;
}

/* ann.cpp:84-93 ClassificationMethod::getThreshold: the value at rank (int)(n*rate). */
static int cmp_float(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}
float orc_get_threshold(const float* dists, int n, float false_accept_rate) {
    float* v = (float*)malloc(sizeof(float) * (size_t)n);
    memcpy(v, dists, sizeof(float) * (size_t)n);
    qsort(v, (size_t)n, sizeof(float), cmp_float);
    int ind = (int)(n * false_accept_rate);
    float t = v[ind];
    free(v);
    return t;
}
