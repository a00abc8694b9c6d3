/*
 * fir_amd.h -- C ABI of the MI355X (gfx950) gallery matcher.
 *
 * Drop-in boundary for the brute-force / probabilistic match path of
 * av-savchenko/fast-image-recognition (qt_cpp). The reference has no FFI of its own -- the path
 * is plain C++ called in-process -- so each entry point below names the reference function it
 * replaces (path:line under the reference checkout); the C++ host shim in
 * fast-image-recognition_amd/host/ re-creates the reference's classes on top of these calls.
 *
 * Conventions
 *   - plain C types only; `fir_gallery` is an opaque handle; no exceptions cross the boundary;
 *   - every function returns FIR_OK (0) or a negative FIR_ERR_* code; fir_last_error() gives
 *     the message of the calling thread's last failure;
 *   - "not found" is reported the reference's way: index -1 and distance 100000
 *     (qt_cpp/db_features.cpp:322-323, qt_cpp/ann.cpp:115-116);
 *   - the gallery rows are COPIED at create time (the reference only borrows them,
 *     qt_cpp/ImageTesting.cpp:40, qt_cpp/ann.h:28): the caller may free its rows afterwards;
 *   - one handle is used by one host thread at a time (the reference is single threaded);
 *   - `*_dev` variants take DEVICE pointers and a HIP stream (void* = hipStream_t, NULL = the
 *     handle's own stream) and are asynchronous; the others take host pointers and return
 *     when the results are in the caller's buffers;
 *   - there is no CPU fallback: without a usable gfx950 device every call fails.
 */
#ifndef FIR_AMD_H
#define FIR_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct fir_gallery fir_gallery;

/* The reference's compile-time metric switch (qt_cpp/db_features.h:12,
 * qt_cpp/db_features.cpp:25-39) as a runtime value. */
enum {
    FIR_METRIC_L2 = 0,   /* sum (a-b)^2 / n                          db_features.cpp:26,40  */
    FIR_METRIC_CHI2 = 1, /* sum_{a+b>0} (a-b)^2/(a+b) / n            db_features.cpp:29-31  */
    FIR_METRIC_KL = 2    /* sum a ln(2a/(a+b)) + b ln(2b/(a+b)) / n  db_features.cpp:33-36  */
};

enum {
    FIR_OK = 0,
    FIR_ERR_ARG = -1,      /* bad argument                                   */
    FIR_ERR_HIP = -2,      /* a HIP runtime call failed                      */
    FIR_ERR_NOMEM = -3,    /* host or device allocation failed               */
    FIR_ERR_NODEVICE = -4, /* no gfx950 device / device index out of range   */
    FIR_ERR_STATE = -5,    /* call not valid for this handle (e.g. no labels) */
    FIR_ERR_COMM = -6      /* an RCCL call failed                             */
};

#define FIR_NOT_FOUND_DIST 100000.0f
#define FIR_KEY_NONE 0xFFFFFFFFFFFFFFFFull

const char* fir_last_error(void);
int fir_version(void);

/* Number of visible HIP devices (does not initialise a device context). */
int fir_device_count(void);
/* name[cap] <- gcnArchName; *cus, *hbm_bytes as reported by the runtime. */
int fir_device_info(int32_t device, char* name, int32_t cap, int32_t* cus, int64_t* hbm_bytes);
/* Spec peak HBM bandwidth in GB/s (the denominator of the roofline report): 8000 on gfx950 (MI350X / MI355X). */
int fir_device_peak_hbm_gbs(int32_t device, double* gbs);

/* ---- gallery ------------------------------------------------------------------------------
 * Replaces the `std::vector<ImageInfo>` gallery that Classifier::train (ImageTesting.cpp:40)
 * and ClassificationMethod's constructor (ann.h:11) borrow. rows[n][d] float32 row-major,
 * class_no[n] (may be NULL: then class lookups fail with FIR_ERR_STATE). */
int fir_gallery_create(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric,
                       int32_t device, fir_gallery** out);
/* Same, rows/class_no already in this device's memory (row-major); async on `stream`. */
int fir_gallery_create_dev(const float* d_rows, int64_t n, int32_t d, const int32_t* d_class_no, int32_t metric,
                           int32_t device, void* stream, fir_gallery** out);
int fir_gallery_destroy(fir_gallery* g);
int fir_gallery_info(const fir_gallery* g, int64_t* n, int32_t* d, int32_t* metric, int32_t* device);
int fir_gallery_set_metric(fir_gallery* g, int32_t metric);
/* Row-sharded galleries: global index of this shard's row 0 (added to every reported index). */
int fir_gallery_set_row_offset(fir_gallery* g, int64_t first_global_row);

/* ---- single pair ----------------------------------------------------------------------------
 * feature_distance(lhs, rhs, start_pos, end_pos), db_features.cpp:22-42 (ImageInfo::distance,
 * db_features.h:24). len = number of floats in each vector. Computed on the device. */
int fir_feature_distance(const float* lhs, const float* rhs, int32_t len, int32_t start_pos, int32_t end_pos,
                         int32_t metric, int32_t device, float* out);

/* ---- nearest row ---------------------------------------------------------------------------
 * Batched recognize_image_bf (db_features.cpp:319-335) / BruteForce::recognize
 * (ann.cpp:113-126): for each of qb queries, the row with the smallest distance over features
 * [start_pos, end_pos) -- first minimum in row order, strict `<` from 100000. end_pos = 0 means d
 * (the reference's max_features == 0 rule, db_features.cpp:320-321).
 * idx[qb] <- row index (+ row offset) or -1, dist[qb] <- its distance (either may be NULL). */
int fir_search_top1(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                    int32_t* idx, float* dist);

/* Device-resident form. d_keys[qb] receives one packed key per query:
 *   (orderable(float bits of distance) << 32) | uint32(row index + row offset),
 * FIR_KEY_NONE when no row qualifies. Keys order exactly like (distance, index), so the minimum
 * of the keys of several row shards is the reference's answer on the whole gallery: reduce them
 * with an integer MIN (RCCL all-reduce on uint64/int64), then unpack. */
int fir_search_top1_keys_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                             uint64_t* d_keys, void* stream);
/* Two batch forms synchronise `stream` once before returning (they have to know that their certificate / candidate lists
 * held, and re-run the exact scan for what did not): L2 whole-range batches that take the matrix-core path (see
 * fir_gallery_set_large_batch_mfma) and chi-square batches of >= 8 queries over >= 65536 rows, which take a nomination scan
 * (1-ulp reciprocal, threshold widened by its error bound) + exact re-rank of the appended rows. The keys are the exact
 * scan's either way; fir_gallery_set_tuning(g, 8, 0) pins the exact scan. */
/* Host-side unpack of packed keys (pure integer work, no device). */
int fir_keys_unpack(const uint64_t* keys, int32_t n, int32_t* idx, float* dist);
uint64_t fir_key_pack(float dist, int32_t idx);
/* classNo of each index (BruteForceClassifier::recognize, ImageTesting.cpp:61-67): -1 stays -1.
 * idx are LOCAL+offset indices as returned above. */
int fir_gallery_classes_of(fir_gallery* g, const int32_t* idx, int32_t n, int32_t* class_out);

/* ---- K nearest rows -------------------------------------------------------------------------
 * The K smallest entries of the distance vector of db_features.cpp:325-333, ascending, equal
 * distances by ascending row index; unused slots idx -1 / dist 100000. idx[qb*k], dist[qb*k]. */
int fir_search_topk(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t k,
                    int32_t* idx, float* dist);
/* Batches of >= 8 queries over >= 65536 rows (L2, whole-chunk ranges) take a candidate-list form: a threshold from a
 * row sample, an append scan at the speed of the top-1 scan, the K smallest of each list -- the same keys; that form
 * synchronises `stream` once before returning (it has to know that no list overflowed; if one did, the register-list
 * scan answers instead). */
int fir_search_topk_keys_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                             int32_t k, uint64_t* d_keys /* [qb*k] ascending */, void* stream);

/* ---- all distances of a feature sub-range ---------------------------------------------------
 * out[qb][n] <- distance(query, row j) over [start_pos,end_pos): the per-row loops of
 * ConventionalTWDClassifier / ProposedTWDClassifier (ImageTesting.cpp:117,174,243). */
int fir_range_distances(fir_gallery* g, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                        float* out);
int fir_range_distances_dev(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                            float* d_out, void* stream);

/* ---- three-way-decision classifiers -----------------------------------------------------------
 * ConventionalTWDClassifier::recognize, ImageTesting.cpp:108-186. type 0 = Posteriors
 * (exp(-100 d) per class, best / sum of the 5 largest class posteriors > threshold, :118-122,141-149),
 * 1 = DistDiff (:157-159), 2 = DistRatio (:160-162). First stage over features
 * [0, reduced_features_count); unreliable queries get the second stage over [.., 256) (:165-180).
 * The gallery needs class labels and d >= 256. class_out[qb] <- class id or -1,
 * unreliable_out[qb] (may be NULL) <- 1 when the second stage ran (num_of_unreliable, :167). */
int fir_twd_conventional(fir_gallery* g, const float* queries, int32_t qb, int32_t num_classes, int32_t type,
                         double threshold, int32_t reduced_features_count, int32_t* class_out, int32_t* unreliable_out);
/* ProposedTWDClassifier::recognize, ImageTesting.cpp:207-288 (CHECK_ALL_INSTANCES build):
 * chunks of reduced_features_count features up to 256, running per-row sums, rows further than
 * best * (1 / threshold) dropped, stop when one class is left. chunks_out (may be NULL) <- chunks used. */
int fir_twd_proposed(fir_gallery* g, const float* queries, int32_t qb, int32_t reduced_features_count, double threshold,
                     int32_t* class_out, int32_t* unreliable_out, int32_t* chunks_out);

/* ---- double-precision classifiers (qt_cpp/classification.cpp) ---------------------------------
 * The training set of KNNClassifier / PNNClassifier (classification.cpp:116-226): train_rows[nt][d]
 * float64 in the reference's scan order -- class 0's training rows, then class 1's, ...
 * (classification.cpp:121-122,195-198), train_class[nt] non-decreasing in [0, num_classes),
 * avg[d] = avgValues (classification.cpp:984; Classifier::normalize subtracts it, :103-105).
 * Rows are copied. */
typedef struct fir_cls fir_cls;
int fir_cls_create(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                   const double* avg, int32_t device, fir_cls** out);
/* Same, train_rows already in `device`'s memory (row-major float64); train_class and avg are host arrays. */
int fir_cls_create_dev(const double* d_train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                       const double* avg, int32_t device, fir_cls** out);
int fir_cls_destroy(fir_cls* c);
/* HIP event pairs around the launches of the float64 distance scan (on the handle's stream); fir_cls_profile_read waits for
 * them and returns their durations (ms) since the previous read, the algorithmic bytes of the last one (training rows once per
 * tile of eight queries + the query tiles + the sums written) and the kernel's name. */
int fir_cls_profile_enable(fir_cls* c, int32_t on);
int fir_cls_profile_read(fir_cls* c, float* ms, int32_t cap, int32_t* count, double* bytes_per_launch, char* kernel, int32_t kernel_cap);
/* PNNwithClusteringClassifier::predict (classification.cpp:389-428) runs the PNN over the medoid rows
 * only but still divides by the FULL training size: set it here (0 = the number of rows held). */
int fir_cls_set_total_training_size(fir_cls* c, int64_t total);
/* sums[qb][nt] <- sum_f ((g_f - avg_f) - (q_f - avg_f))^2 per training row, accumulated in feature
 * order in double (classification.cpp:123-141 before the division, :199-211). */
int fir_cls_distance_sums(fir_cls* c, const double* queries, int32_t qb, double* sums);
/* PNNClassifier::predict_bf, classification.cpp:188-226. var <= 0 selects the reference's value
 * (2e-5, divided by 10 when d > 2000, :190-193). scores[qb][num_classes] (may be NULL) <-
 * sum_t exp(-dist / (2 d var)) / nt per class; best_class[qb] <- first maximum. */
int fir_cls_pnn_predict(fir_cls* c, const double* queries, int32_t qb, double var, double* scores, int32_t* best_class);
/* PNNClassifier::predict_sequentional, classification.cpp:228-295: 32-feature chunks, running per-row
 * sums, class outputs with 2*var*max_fi, classes below max/1e9 dropped, stop when one class is left.
 * chunks_out[qb] (may be NULL) <- chunks used. */
int fir_cls_pnn_predict_seq(fir_cls* c, const double* queries, int32_t qb, double var, int32_t* best_class, int32_t* chunks_out);
/* KNNClassifier::predict, classification.cpp:116-170: rows sorted by mean distance vote for their
 * class until one class has k votes. 1 <= k <= 8. best_class[qb].
 * Batches of >= 128 queries against a training set that streams from HBM (> 256 MiB of rows) go through the matrix cores: an fp16
 * copy of the centred rows nominates the nearest rows (one for k = 1, eight for k > 1), the reference's float64 arithmetic re-ranks
 * them and a rounding-error certificate proves that no other row can be among them; the vote is taken over those rows. A query that
 * is not certified, whose eight rows do not settle the vote, or with equal distances among them goes through the exact scan of every
 * row as before: the classes are the exact scan's either way. Costs the fp16 copy (nt * d * 2 bytes), made on first use. */
int fir_cls_knn_predict(fir_cls* c, const double* queries, int32_t qb, int32_t k, int32_t* best_class);
/* min_queries > 0: kNN batches of at least that many queries take the matrix cores whatever the training set's size; 0: never (frees
 * the fp16 copy); < 0: back to the automatic choice above. */
int fir_cls_set_knn_mfma(fir_cls* c, int32_t min_queries);
/* queries that took the matrix-core path so far, and how many of them the exact scan answered after all */
int fir_cls_knn_stats(fir_cls* c, int64_t* matrix_core_queries, int64_t* exact_scan_queries_of_them);
/* the dominant kernel of the most recent PROFILED launch (fir_cls_profile_enable): name, algorithmic bytes (scans), dot-product flops
 * (matrix-core passes: 2 * rows * d * queries of the launch) */
int fir_cls_last_dispatch(fir_cls* c, char* kernel, int32_t kernel_cap, double* bytes_per_launch, double* flops_per_launch);
/* The part of that vote a row shard can do alone: nearest[qb][num_classes][k] <- the k smallest mean distances of every
 * class among the rows held, ascending, DBL_MAX where the class has fewer. The class that first collects k votes in the
 * globally sorted order is the one whose k-th nearest member is nearest, so ranks exchange these lists, keep the k
 * smallest per class and take the first minimum of the k-th values (sharding.py: merge_knn_class_nearest). */
int fir_cls_knn_class_nearest(fir_cls* c, const double* queries, int32_t qb, int32_t k, double* nearest);

/* ---- FPNNClassifier: orthogonal-series (trigonometric) PNN, classification.cpp:618-791 ----------
 * train() (:661-696): train_rows[nt][d] float64, class-major like fir_cls_create (train_class non-decreasing), avg[d] /
 * sd[d] = avgValues / stdValues of split_train_test (:969-989), scale = features_scale. J = max(3, ceil(cbrt(nt /
 * num_classes))) harmonics; the model has d * num_classes * (2J+1) doubles. */
typedef struct fir_fpnn fir_fpnn;
int fir_fpnn_train(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                   const double* avg, const double* sd, double scale, int32_t device, fir_fpnn** out);
int fir_fpnn_destroy(fir_fpnn* h);
int fir_fpnn_info(const fir_fpnn* h, int32_t* J, int32_t* d, int32_t* num_classes);
/* a_out[(f * num_classes + c) * (2J+1) + k]: the reference's `a` (:676-690). */
int fir_fpnn_get_model(fir_fpnn* h, double* a_out);
/* predict_bf (:698-735): best_class[qb]; outputs[qb][num_classes] (may be NULL) = the float log-scores. */
int fir_fpnn_predict(fir_fpnn* h, const double* queries, int32_t qb, int32_t* best_class, float* outputs);
/* predict_sequentional (:736-791): 32-feature chunks, classes below max + fastlog(output_ratio) * features_seen are
 * dropped, stop when one is left. chunks_out[qb] (may be NULL) = chunks evaluated. */
int fir_fpnn_predict_seq(fir_fpnn* h, const double* queries, int32_t qb, float output_ratio, int32_t* best_class,
                         int32_t* chunks_out);

/* ---- DirectedEnumeration (maximum-likelihood directed enumeration, ann.h:64-100) -------------------
 * The PIVOT build of the constructor, ann.cpp:302-331: for ii = 0..n_pivots-1, table row ii = distance(gallery row j,
 * pivot ii) over all d features (ann.h:33-38), min_other[ii] = smallest distance from pivot ii to a row of another class
 * (:309-311,325 -> otherClassesDists, the input of getThreshold :341-343), and pivot ii+1 = the first row whose summed
 * distance to the pivots so far (-1000000 restart at a pivot, :313-318) is largest and > 0 (:319-322).
 * pivots[0] = first_pivot (the reference draws it with random_shuffle, :366-376). The gallery needs class labels.
 * pivots_out[n_pivots], table_out[n_pivots][n] (may be NULL), min_other_out[n_pivots] (may be NULL).
 * *n_built_out (may be NULL) < n_pivots when no row had a positive sum (identical rows): the reference would index
 * dbImages[-1] there; the remaining pivots are -1 and their table rows unspecified. */
int fir_dem_pivot_table(fir_gallery* g, int32_t first_pivot, int32_t n_pivots, int32_t* pivots_out, float* table_out,
                        float* min_other_out, int32_t* n_built_out);

/* The same build kept on the device for query time: the table rows of the first min(n_built, 32) pivots
 * (ann.cpp:333-334 keeps 32) and those pivots as a small gallery of their own. The gallery handle must outlive it. */
typedef struct fir_dem fir_dem;
int fir_dem_create(fir_gallery* g, int32_t first_pivot, int32_t n_pivots, fir_dem** out);
int fir_dem_destroy(fir_dem* h);
int fir_dem_info(const fir_dem* h, int32_t* n_pivots, int32_t* n_built, int32_t* n_used, int64_t* n);
/* pivots_out[n_pivots], min_other_out[n_pivots], table_out[n_used][n], order_out[n] = the state of
 * `likelihood_indices` after the pivot loop of recognize (ann.cpp:427-432; identity except near the front). Any may be NULL. */
int fir_dem_get(fir_dem* h, int32_t* pivots_out, float* min_other_out, float* table_out, int32_t* order_out);
/* The data-parallel part of DirectedEnumeration::recognize (ann.cpp:427-447) for qb queries:
 * pivot_dist_out[qb][n_used] = distance(query, pivot i) (CHECK_FOR_BEST_DIST's tmpDist), and
 * lik_out[qb][n] = `likelihoods` after all n_used pivots: sum over i, in order, of (tmpDist_i - table[i][nu])^2 in
 * float, each row visited as often as the reference's index bookkeeping visits it. Either may be NULL. */
int fir_dem_likelihoods(fir_dem* h, const float* queries, int32_t qb, float* pivot_dist_out, float* lik_out);

/* out[qb][m] = distance(query q, gallery row rows[q][m]) over [start_pos, end_pos) (end_pos 0 = d): the candidate
 * checks of an enumeration (CHECK_FOR_BEST_DIST, ann.cpp:389-399) as one gather. Rows are LOCAL indices; a row
 * outside the gallery yields 100000. */
int fir_rows_distances(fir_gallery* g, const float* queries, int32_t qb, const int32_t* rows, int32_t m, int32_t start_pos,
                       int32_t end_pos, float* out);

/* ---- large query batches through the matrix cores (L2, whole feature range) ------------------------
 * Same answers as fir_search_top1 -- bit-identical index and distance: an MFMA GEMM only nominates
 * candidate rows, the reference's arithmetic re-ranks them, a rounding-error certificate proves no other
 * row can win, and uncertified queries are re-run through the exact streaming scan (fir_gemm.hip).
 * Costs one extra copy of the gallery in MFMA fragment order. The gallery handle must outlive it. */
typedef struct fir_gemm fir_gemm;
enum {
    FIR_GEMM_F32 = 0,        /* v_mfma_f32_32x32x2_f32: exact products (157 TF peak)                                */
    FIR_GEMM_BF16_SPLIT = 1, /* x = hi + lo in bf16, hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16           */
    FIR_GEMM_F16 = 2         /* (default) one v_mfma_f32_32x32x16_f16 term on power-of-two-scaled fp16 copies: half the gallery
                              * bytes and a third of the MFMAs per 128 queries; the proxy is good to 2^-10 |q||g|, which the
                              * certificate carries: more rows fall inside the rounding window and are re-ranked exactly */
};
int fir_gemm_create(fir_gallery* g, fir_gemm** out);   /* = fir_gemm_create_ex(g, FIR_GEMM_F16, out) */
int fir_gemm_create_ex(fir_gallery* g, int32_t precision, fir_gemm** out);
/* ... over a feature prefix [0, end_pos) of every row (0 or d: the whole row): end_pos a multiple of 16, FIR_GEMM_F16 only.
 * The reference's "BF, 64" / "BF, 256" classifiers (ImageTesting.cpp:526-529) compare prefixes; the search entry points
 * route such batches here by themselves (start_pos == 0, end_pos % 16 == 0, end_pos >= 64, same thresholds). */
int fir_gemm_create_range(fir_gallery* g, int32_t precision, int32_t end_pos, fir_gemm** out);
int fir_gemm_destroy(fir_gemm* m);
int fir_gemm_search_top1_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, uint64_t* d_keys, void* stream);
/* 1..8 queries: one pass over the fp16 copy with v_dot2 (no matrix cores), the rows within one rounding window of the smallest proxy of
 * ALL rows re-ranked exactly, same certificate, same keys. Half the bytes of the exact scan's pass: the search entry points route
 * ONE-query L2 calls here when the compared rows are >= 1.5 GB, from the gallery's 17th such call on (332 -> 253 us for one query against 1M x 512; with two queries the
 * forms tie, beyond that the f32 scan is faster). */
int fir_gemm_search_few_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, uint64_t* d_keys, void* stream);
/* The k nearest rows (2 <= k <= 8; k = 1 is the call above): d_keys[q * k + r], ascending, the keys fir_search_topk_keys_dev
 * gives. The bound of the append pass is an order statistic of a row sample plus one rounding window; the re-rank window hangs
 * on the k-th smallest proxy and the certificate is taken against the k-th exact distance. fir_search_topk[_keys_dev] route
 * whole-range L2 batches here under the same rule as the top-1 calls (fir_gallery_set_large_batch_mfma). */
int fir_gemm_search_topk_keys_dev(fir_gemm* m, const float* d_queries, int32_t qb, int32_t k, uint64_t* d_keys, void* stream);
/* fir_search_top1 and fir_search_top1_keys_dev send L2 whole-range batches through this path BY DEFAULT when the batch
 * has >= 128 queries and the gallery >= 65536 rows (smaller, cache-resident galleries: where a cost model of the two forms gives the
 * matrix cores 15 % or more -- e.g. 128 queries against 40 000 x 512, 1 024 against 8 192 x 512 --, from the gallery's fourth such
 * call on; fewer queries on larger galleries: 32 from 128 MB of compared rows on; from 300 MB on wherever a cost model of the two forms says so -- 3, 5, 6, 7 queries are two or three scan passes -- down to 3 queries) (created on first use; costs the extra fp16 gallery copy, n*d*2
 * bytes; identical keys). min_queries > 0: the caller's threshold instead (any gallery size); 0: never, frees the copy
 * (the exact streaming scan answers everything); < 0: back to the default. */
int fir_gallery_set_large_batch_mfma(fir_gallery* g, int32_t min_queries);
/* passes = 64-query GEMM passes queued so far, fallback_queries = queries answered by the exact device scan instead (results
 * identical either way). Uncertified queries are dealt with on the device, in stream order (a second matrix-core pass with the
 * tightest bound the first one justifies, then the exact scan): the search calls never synchronise; this call waits for the
 * device and reads the counters. */
int fir_gemm_stats(const fir_gemm* m, int64_t* passes, int64_t* fallback_queries);
/* out[0] = passes, out[1] = queries whose FIRST certificate did not hold (list overflow, window reaching the bound, NaN),
 * out[2] = queries the exact device scan answered (= fallback_queries above). */
int fir_gemm_stats_ex(const fir_gemm* m, int64_t out[3]);
/* Diagnostics: the first (up to 8) queries of this state's life whose first certificate did not hold -- per query four floats:
 * entries its candidate list was asked to take (4096 fit), the bound the pass appended below, the smallest stored proxy (K-th
 * smallest for the K nearest), |q|^2. *count = filled slots. Waits for the device. */
int fir_gemm_uncertified_notes(const fir_gemm* m, float out[32], int32_t* count);
int fir_gallery_mfma_uncertified_notes(fir_gallery* g, float out[32], int32_t* count);   /* ... of the automatic dispatch's whole-row state */
/* The same counters for the matrix-core states a gallery's AUTOMATIC dispatch has built (whole rows and feature prefixes),
 * summed: fallback_queries = queries it could not certify and sent through the exact scan (results identical either way). */
int fir_gallery_mfma_stats(fir_gallery* g, int64_t* passes, int64_t* fallback_queries);
int fir_gallery_mfma_stats_ex(fir_gallery* g, int64_t out[3]);
/* HBM held by a gallery handle, in bytes: the tiled f32 rows (+ labels), the fp16 fragment copies the automatic dispatch has
 * made (0.5 x the rows each; one per feature prefix in use), the row-major f32 shadow copies the exact re-rank gathers from
 * (1 x the rows; made only when four times their size was free), and all other scratch. Any pointer may be NULL. */
int fir_gallery_memory_bytes(fir_gallery* g, int64_t* tiled, int64_t* fp16_fragments, int64_t* rowmajor_shadow, int64_t* scratch);
/* Which copies of the gallery the automatic dispatch may keep next to the tiled rows: FIR_SHADOW_ALL (default), FIR_SHADOW_FP16
 * (fragments only: the re-rank gathers from the tiled rows), FIR_SHADOW_NONE (none: every call takes the exact scan, like
 * fir_gallery_set_large_batch_mfma(g, 0)). Takes effect for states built afterwards; existing ones are dropped. */
enum { FIR_SHADOW_NONE = 0, FIR_SHADOW_FP16 = 1, FIR_SHADOW_ALL = 2 };
int fir_gallery_set_shadow_copies(fir_gallery* g, int32_t mode);

/* ---- one gallery sharded by rows over several GPUs (BASELINE.json configs[3]) -----------------
 * The reference is single threaded and single device; this is how BruteForce::recognize (ann.cpp:113-126) and
 * BruteForceClassifier::recognize (ImageTesting.cpp:58-71) use a whole node. Rows are split into contiguous blocks
 * of whole 64-row tiles, one block (or shards_per_device blocks) per listed device; every shard scans the batch and
 * the ranks exchange
 *   top-1: ncclAllReduce(ncclMin, ncclUint64) over the packed keys -- the low word is the GLOBAL row index, so the
 *          integer minimum is the first-minimum rule of db_features.cpp:329-332 over the whole gallery;
 *   top-K: ncclAllGather of K keys per query per rank + an integer K-way merge;
 *   class: ncclAllReduce(ncclMin, ncclInt32) of classNo-if-I-hold-the-winning-row.
 * RCCL over xGMI; the handle owns the per-device streams and communicators. Results are identical to the
 * one-device calls on the unsplit gallery (index, distance bits, tie-break). */
typedef struct fir_sharded fir_sharded;
#define FIR_COMM_ID_BYTES 128
typedef struct fir_shard_opts {
    int32_t struct_bytes;       /* sizeof(fir_shard_opts) */
    int32_t shards_per_device;  /* logical shards per listed device (0 or 1: one); > 1 exercises the split on few GPUs   */
    int64_t first_global_row;   /* global index of rows[0]: this process's row block in a multi-process gallery          */
    const void* comm_id;        /* NULL: this process holds the whole gallery. Else FIR_COMM_ID_BYTES bytes that process 0
                                 * got from fir_comm_unique_id and handed to every process (any out-of-band channel)     */
    int32_t proc_rank, nprocs;  /* with comm_id: this process among nprocs; each lists the same NUMBER of devices        */
    int32_t rows_on_device;     /* 1: rows / class_no are device pointers on devices[0] (one-entry device list only)     */
    int32_t reserved;
    int64_t total_rows;         /* fir_cls_create_sharded with comm_id: training rows over ALL processes (PNN divisor); 0 = n */
    int32_t timeout_ms;         /* bound of every wait behind a collective (0: 120 000). When it passes, or RCCL reports an
                                 * asynchronous error, the communicator is aborted and the call returns FIR_ERR_COMM          */
    int32_t fail_shard;         /* AUDIT BUILD ONLY (libfir_amd_audit.so, -DFIR_AUDIT; the shipped library returns FIR_ERR_ARG for a
                                 * non-zero value): 1-based index of the local shard whose step fails with FIR_ERR_NOMEM (0: none)   */
    int32_t fail_step;          /* ... 1: its scan, after the buffers were agreed on; 2: the buffer growth of its device; 3: from the
                                 * handle's second exchange on the status element comes back poisoned, as if a PEER's scan had failed */
    int32_t reserved2;
} fir_shard_opts;
/* Failure semantics of every sharded handle (the reference's convention is "-1, never block", ann.cpp:113-126): a call in which
 * ANY rank fails -- an allocation, a scan, an RCCL call, a peer that never shows up within timeout_ms -- returns a negative
 * FIR_ERR_* on EVERY rank and never blocks for longer than the time-out: buffers grow in a step the ranks agree on (one-int
 * ncclAllReduce), a rank whose scans fail still enters the exchange with neutral keys and a poisoned status element, waits are
 * bounded and poll ncclCommGetAsyncError, a broken communicator is aborted. The handle is then closed: further calls return
 * FIR_ERR_STATE at once; fir_sharded_destroy / fir_cls_sharded_destroy free it and a fresh handle works. For the asynchronous
 * device-pointer call the failing rank gets its error from the call itself, the others from fir_sharded_sync. */
int fir_comm_unique_id(void* id_out /* [FIR_COMM_ID_BYTES] */);
/* Single process, all rows, one shard per listed device. devices[ndev]: HIP device indices, no repeats. */
int fir_gallery_create_sharded(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric,
                               const int32_t* devices, int32_t ndev, fir_sharded** out);
/* General form. Collective over all processes of the communicator (like every search call below). */
int fir_gallery_create_sharded_ex(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric,
                                  const int32_t* devices, int32_t ndev, const fir_shard_opts* opts, fir_sharded** out);
int fir_sharded_destroy(fir_sharded* h);
int fir_sharded_info(const fir_sharded* h, int64_t* n_local, int32_t* d, int32_t* ndev, int32_t* nshards, int32_t* nranks,
                     int32_t* first_rank);
/* Shard i of this process (borrowed: tuning, profiling; NULL when the shard holds no rows), its first global row, its rows. */
int fir_sharded_shard(fir_sharded* h, int32_t i, fir_gallery** g, int64_t* first_global_row, int64_t* rows);
int fir_sharded_set_metric(fir_sharded* h, int32_t metric);
/* fir_search_top1 / fir_search_topk over the whole sharded gallery; indices are global rows. */
int fir_sharded_search_top1(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                            int32_t* idx, float* dist);
int fir_sharded_search_topk(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                            int32_t k, int32_t* idx, float* dist);
/* Batched BruteForceClassifier::recognize (ImageTesting.cpp:58-71): class_out[qb] <- classNo of the nearest row or -1;
 * idx / dist may be NULL. Needs class labels. */
int fir_sharded_classify_top1(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                              int32_t* class_out, int32_t* idx, float* dist);
/* One process per GPU (one-entry device list): device pointers on that GPU, asynchronous on `stream`
 * (NULL = the handle's); d_keys[qb] <- the reduced keys, identical on every rank.
 * A rank whose own scans fail gets its error from the call; its peers learn of it from the status element that travels
 * behind the keys: it lands in a sticky word that (a) the NEXT call on the handle looks at on entry, if the previous call's
 * work is through by then, and (b) fir_sharded_sync waits for -- on the stream the call actually ran on, with the handle's
 * time-out -- and reports. A peer that failed skips every later collective, so a caller that must never outrun a failure
 * calls fir_sharded_sync between asynchronous calls (or at least before it relies on the keys); without it the second of two
 * back-to-back calls can enqueue a collective the failed peer never enters, which then ends at the time-out of the next sync. */
int fir_sharded_search_top1_keys_dev(fir_sharded* h, const float* d_queries, int32_t qb, int32_t start_pos,
                                     int32_t end_pos, uint64_t* d_keys, void* stream);
/* Waits (bounded by timeout_ms) for everything the handle has queued, the asynchronous calls on their callers' streams included,
 * and returns the worst status any of their exchanges came back with (FIR_OK, or the peer's error: the handle is then closed). */
int fir_sharded_sync(fir_sharded* h);

/* PNNClassifier::predict_bf (classification.cpp:188-226) over training rows split across GPUs: every shard sums
 * exp(-dist / (2 d var)) over ITS rows per class with the GLOBAL training-set size as divisor, the partial class scores
 * are added on the device and across ranks with ncclAllReduce(ncclSum, ncclDouble), then the first maximum is the class.
 * Only the order of the additions differs from the one-device call (scores within 1e-12 relative, same arg-max unless
 * two classes tie to that precision). train_rows / train_class as in fir_cls_create (class-major); shards are
 * contiguous row blocks. opts may be NULL (one process, one shard per device). */
typedef struct fir_cls_sharded fir_cls_sharded;
int fir_cls_create_sharded(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes,
                           const double* avg, const int32_t* devices, int32_t ndev, const fir_shard_opts* opts,
                           fir_cls_sharded** out);
int fir_cls_sharded_destroy(fir_cls_sharded* h);
int fir_cls_sharded_pnn_predict(fir_cls_sharded* h, const double* queries, int32_t qb, double var, double* scores,
                                int32_t* best_class);
/* KNNClassifier::predict (classification.cpp:116-170) over the same shards: every shard's k smallest mean distances per class
 * (fir_cls_knn_class_nearest) are merged on the device and across ranks (ncclAllGather); the class whose k-th nearest member is
 * nearest is the one that first collects k votes in the globally sorted order. Exact: the same class as the one-device call. */
int fir_cls_sharded_knn_predict(fir_cls_sharded* h, const double* queries, int32_t qb, int32_t k, int32_t* best_class);
/* HIP events around the exchange step on this process's first device: durations (ms) since the previous read. */
int fir_sharded_profile_enable(fir_sharded* h, int32_t on);
int fir_sharded_profile_read(fir_sharded* h, float* exchange_ms, int32_t cap, int32_t* count);

/* ---- profiling ------------------------------------------------------------------------------
 * When enabled, every gallery-scan kernel launch is bracketed by HIP events on the stream it
 * is launched on. fir_profile_read waits for them and returns the launch durations (ms) in
 * launch order since the previous read; *bytes_per_launch is the algorithmic byte count of the
 * LAST launch (gallery range + query tile + keys). */
int fir_profile_enable(fir_gallery* g, int32_t on);
int fir_profile_read(fir_gallery* g, float* ms, int32_t cap, int32_t* count, double* bytes_per_launch);
/* What the most recent top-1 search on this handle dispatched: its dominant kernel (the gallery-streaming one) as the
 * library launched it, with the code object's resource numbers -- so that a benchmark reports the kernel that ran,
 * not the one it expects. With profiling on, fir_profile_read's durations are the launches of exactly this kernel. */
typedef struct fir_dispatch_info {
    int32_t struct_bytes;     /* in: sizeof(fir_dispatch_info) */
    int32_t path;             /* 0 = exact streaming scan; 1 = matrix-core nomination + exact re-rank + certificate */
    char kernel[160];         /* e.g. "fir::k_scan_l2_lds<1, 8, 4, false>" */
    int32_t launches;         /* launches of it in that call */
    int32_t grid_x, grid_y, block;
    int32_t lds_bytes;        /* static + dynamic LDS of one workgroup */
    int32_t vgprs;            /* registers per lane (hipFuncGetAttributes) */
    int32_t queries_per_pass; /* queries that share one read of the gallery */
    double bytes_per_launch;  /* algorithmic HBM bytes of one launch */
    double flops_per_launch;  /* matrix-core path: 2 * rows * d * queries of one launch; 0 for the scan */
    int32_t warmup_calls_left; /* automatic dispatch, small-saving cases (cache-resident galleries, one-query calls): the matrix-core
                               * state is built only for a gallery that keeps getting such calls; > 0 = this call was one of the
                               * first few and took the scan, this many more will; 0 = steady state */
    int32_t reserved;
    char knobs[160];          /* the FIR_* environment knobs this PROCESS has honoured so far, space separated ("" = none; a trailing
                               * "..." = more than fit). Experiment switches only: none of them changes an answer in the shipped library
                               * (the audit knobs that can are compiled into libfir_amd_audit.so alone, which also lists them here) */
} fir_dispatch_info;
int fir_gallery_last_dispatch(fir_gallery* g, fir_dispatch_info* out);

/* Chi-square / KL galleries: which division sequence the scans use. The IEEE f32 division the compiler emits carries
 * scaling and fix-up steps for operands near the ends of the exponent range; when every gallery value and every query
 * value of a call is +0 or within [2^-26, 2^16] (any L1- or L2-normalised non-negative feature vector is) those steps
 * are the identity, and the scans run the same arithmetic without them -- the same bits, about half the VALU work for
 * chi-square and a third for KL. The check is made on the device while rows are uploaded and queries transposed; one
 * value outside the range (a negative, a NaN, a denormal) and the whole call takes the full sequence.
 * *gallery_plain: every uploaded value was in range; *last_queries_plain: so was every query value of the most recent
 * search on this handle. Synchronises the handle's stream. */
int fir_gallery_value_range(fir_gallery* g, int32_t* gallery_plain, int32_t* last_queries_plain);

/* Block until all work queued by this handle is done. */
int fir_gallery_sync(fir_gallery* g);

/* Tunables (for experiments). queries_per_pass: 1, 2, 4, 8 or 16; 0 keeps the current setting, < 0 returns to the
 * automatic choice: at most 8 for galleries streamed from HBM and 16 for galleries that stay resident in the 256 MiB
 * Infinity Cache (8 when a 16-query tile would not fit 64 KiB of LDS, d > 1020), halved per call until
 * tiles x passes gives every SIMD a wave (small galleries, small batches). waves: waves per gallery pass, 0 = automatic. */
int fir_gallery_set_tuning(fir_gallery* g, int32_t queries_per_pass, int32_t waves);
int fir_gallery_get_tuning(const fir_gallery* g, int32_t* queries_per_pass, int32_t* waves, int32_t* max_waves);

#ifdef __cplusplus
}
#endif
#endif /* FIR_AMD_H */
