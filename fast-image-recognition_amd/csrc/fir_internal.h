// fir_internal.h -- what the library's translation units share (not part of the public ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct fir_gallery;

// Read-only view of a gallery handle for the other translation units (fir_twd.hip).
struct fir_gallery_view {
    int device;
    int cus;
    int64_t n;
    int d;
    int metric;             // FIR_METRIC_*
    int64_t row_offset;
    const int32_t* cls;     // device, may be NULL
    hipStream_t stream;     // the handle's own stream
};
extern "C" int fir_gallery_view_(fir_gallery* g, fir_gallery_view* out);
extern "C" void fir_set_last_error_(const char* msg);
// Every environment knob the library honours goes through here: getenv(name), and a set one is remembered (once) in the list
// fir_gallery_last_dispatch reports in fir_dispatch_info::knobs -- a stray variable in a production environment is visible.
// The knobs that can change ANSWERS (FIR_GEMM_EREL_SCALE, FIR_GEMM_DBG_SKIP, FIR_GEMM_ADAPT_DBG, fir_shard_opts.fail_*) exist
// only in the audit build (-DFIR_AUDIT, libfir_amd_audit.so: what the tests that need them load); the shipped library ignores them.
extern "C" const char* fir_knob_(const char* name);
extern "C" int fir_gallery_tiled_(fir_gallery* g, const void** gal4, int* dp4);   // the tiled f32 gallery (fir_kernels.h layout)

// Device scratch owned by the gallery handle: `slot` in [0, 24), grown on demand, kept until the gallery is destroyed
// (the per-call hipMalloc / hipFree pairs of the classifier entry points cost more than their kernels on small galleries).
// Slots 0-7 and 16: fir_twd.hip, 8-11: fir_dem.hip, 12-15: fir_capi.hip.
extern "C" int fir_gallery_scratch_(fir_gallery* g, int slot, size_t bytes, void** out);
// Per-handle call counters of the other translation units (slot 0: fir_twd.hip's fused classifier): returns the value before the increment.
extern "C" uint64_t fir_gallery_next_counter_(fir_gallery* g, int slot);

// d_out[(ci * qb + q) * n + row] = distance(query q, row) over sub-range ci = [start + ci*step, start + (ci+1)*step), for
// every sub-range of [start, end): ONE gallery pass (k_scan_subranges) when step is a multiple of 32 features, one
// range-distance pass per sub-range otherwise. Device pointers; asynchronous on `stream`.
extern "C" int fir_subrange_distances_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t step,
                                           float* d_out, void* stream);

// d_out[q * n + row] = distance over [0, split), d_out[(qb + q) * n + row] = distance over [split, end): ONE gallery pass when
// both widths are multiples of 32 features (the two stages of the conventional TWD: 64 and 192), two otherwise.
extern "C" int fir_split_distances_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t split, int32_t end, float* d_out, void* stream);

// First touch of a device by this library. The HIP runtime's start-up draws from libc's rand(); the reference's harnesses
// lean on that stream (std::random_shuffle in getTrainingAndTestImages and DirectedEnumeration::init, srand(13) in
// testRecognitionMethod), so the start-up runs on a private random state and the caller's is handed back untouched.
extern "C" int fir_runtime_init_(int device);

// Small host-pointer calls of the other translation units, without copy engine and without stream synchronisation: the
// handle's pinned, device-visible buffer (queries go in at `base`, up to *query_bytes; results come back at *results, 4096
// eight-byte words), a fresh ticket number, and the wait for the word a call's last kernel writes it to (spins for 2 ms,
// then synchronises the stream). See fir_search_top1 in fir_capi.hip.
extern "C" int fir_gallery_pin_(fir_gallery* g, void** base, size_t* query_bytes, uint64_t** results);
extern "C" uint64_t fir_gallery_next_ticket_(fir_gallery* g);
extern "C" int fir_gallery_wait_ticket_(fir_gallery* g, volatile uint64_t* flag, uint64_t ticket);

// The exact streaming scan whatever the batch size (fir_search_top1_keys_dev may route large L2 batches through fir_gemm_*,
// whose uncertified queries must not come back to it).
extern "C" int fir_search_top1_exact_keys_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos,
                                               uint64_t* d_keys, void* stream);

// ... and the exact K-nearest-rows form over features [0, end_pos), never through fir_gemm_*.
extern "C" int fir_search_topk_exact_keys_dev_(fir_gallery* g, const float* d_queries, int32_t qb, int32_t end_pos, int32_t k, uint64_t* d_keys, void* stream);

// fir_profile_enable / fir_profile_read / fir_gallery_last_dispatch for kernels launched by the other translation units:
// an event pair around ONE launch on `st` (no-ops unless profiling is on), and the record of the call's dominant kernel.
extern "C" int fir_gallery_profile_begin_(fir_gallery* g, void* st);
extern "C" int fir_gallery_profile_end_(fir_gallery* g, void* st, double bytes_alg);
extern "C" void fir_gallery_note_dispatch_(fir_gallery* g, const void* fn, const char* name, int first, int gx, int gy, int block, size_t dyn_lds,
                                           int qpp, double bytes, double flops);

struct fir_gemm;
// fir_gemm_search_top1 / _topk_keys_dev for queries still in host memory: uploaded into d_stage one super-batch at a time, each
// upload under the previous super-batch's full passes.
extern "C" int fir_gemm_search_staged_(fir_gemm* m, const float* h_queries, float* d_stage, int32_t qb, int32_t k, uint64_t* d_keys, void* stream);

// fir_gemm_create_range with the re-rank's row-major shadow copy decided by the caller: -1 = when HBM has room for it (or as
// FIR_GEMM_ROWMAJOR says), 0 = never, 1 = whenever it can be allocated.
extern "C" int fir_gemm_create_range_ex_(fir_gallery* g, int32_t precision, int32_t end_pos, int32_t rowmajor_mode, fir_gemm** out);
// device bytes one matrix-core state holds: the fp16 (bf16 / f32) fragment copy, the row-major shadow, everything else (scratch)
extern "C" void fir_gemm_memory_bytes_(const fir_gemm* m, int64_t* fragments, int64_t* rowmajor, int64_t* scratch);

struct fir_cls;
extern "C" int fir_cls_pnn_scores_dev_(fir_cls* c, const double* queries, int32_t qb, double var, double** d_scores, void** stream, int32_t* max_batch);
extern "C" int fir_cls_knn_nearest_dev_(fir_cls* c, const double* queries, int32_t qb, int32_t k, double** d_lists, void** stream, int32_t* max_batch);

// fir_gemm_f64.h: the matrix-core nomination over a float64 training set (tiled layout of fir_cls.hip), for large kNN batches
extern "C" int fir_gemm_create_f64_(int device, int cus, void* stream, const void* gal2, int64_t nt, int d, int dp2, fir_gemm** out);
extern "C" int fir_gemm_knn_f64_(fir_gemm* m, const double* d_qc, int32_t qb, int32_t kp, int32_t* d_rows, double* d_dist, int32_t* d_ok, void* stream,
                                 const char** kernel_name, double* flops_per_launch, hipEvent_t* ev_pair);
extern "C" int fir_gemm_destroy(fir_gemm* m);
