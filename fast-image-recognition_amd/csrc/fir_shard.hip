// fir_shard.hip -- one gallery sharded by rows over several GPUs, behind the C ABI.
//
// Every shard is an ordinary fir_gallery on its device with row_offset = the global index of its first row; a query
// batch is scanned by all shards at once and the only exchange is
//   top-1:  ncclAllReduce(ncclMin, ncclUint64) of qb packed keys (the low word is the GLOBAL row index, so the integer
//           minimum is the reference's first-minimum rule over the whole gallery, db_features.cpp:329-332);
//   top-K:  ncclAllGather of every rank's K keys per query + an integer K-way merge;
//   class:  ncclAllReduce(ncclMin, ncclInt32) of "classNo of the winning row if I hold it, else INT32_MAX".
// All payloads are KB-sized: the collectives are latency-bound and run on the stream that produced the keys.
//
// Two ways to span GPUs, freely combined (ranks = processes x devices per process):
//   * one process, a device list: one worker thread per device issues that device's launches and its RCCL call;
//   * one process per GPU (torch.distributed.run style): process 0 makes an id with fir_comm_unique_id, hands it
//     to the others out of band, and every process passes it in fir_shard_opts.
// Several logical shards per device (shards_per_device) exist so that the whole path -- split, per-shard scan,
// on-device minimum, RCCL call -- can be exercised on a one-GPU box.
//
// Failure semantics: the reference's convention is "-1, never block" (ann.cpp:113-126, ImageTesting.cpp:63). A collective
// in which one rank does not take part blocks every other rank for good, so no rank ever leaves a call between two
// collectives of it:
//   * buffers grow in an AGREED step: the rank tries its allocations, then a one-int ncclAllReduce(ncclMin) of the local
//     status (on a word allocated when the communicator was made) tells every rank whether all of them have their
//     buffers; the step runs only in calls that have to grow something, which are the same calls on every rank (the
//     capacities depend on the sequence of batch shapes only);
//   * a rank whose scans fail afterwards still enters the exchange, with neutral keys (FIR_KEY_NONE / +0 sums) and a
//     poisoned status element that travels behind the payload (min / gather / sum like the payload itself): every rank
//     returns the same FIR_ERR_* from that call;
//   * every wait on a stream that carries a collective is bounded (fir_shard_opts.timeout_ms, default 120 s) and polls
//     ncclCommGetAsyncError; an error or the time-out aborts the communicator (ncclCommAbort);
//   * after any such failure the handle is dead: later calls return FIR_ERR_STATE at once on every rank (they have all
//     seen the failure), fir_sharded_destroy still frees it, and a fresh handle works.
// fir_shard_opts.fail_shard / fail_step inject a failure for the tests.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/fir_amd.h"
#include "fir_common.h"
#include "fir_internal.h"

namespace {

using fir::kKeyNone;

thread_local char g_sh_err[512];
int sh_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_sh_err, sizeof(g_sh_err), fmt, ap);
    va_end(ap);
    fir_set_last_error_(g_sh_err);
    return code;
}
#define SH_HIP(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) return sh_fail(e_ == hipErrorOutOfMemory ? FIR_ERR_NOMEM : FIR_ERR_HIP, "%s failed: %s (%s:%d)", #expr, \
                                             hipGetErrorString(e_), __FILE__, __LINE__);                               \
    } while (0)
#define SH_NCCL(expr)                                                                                                  \
    do {                                                                                                               \
        ncclResult_t r_ = (expr);                                                                                      \
        if (r_ != ncclSuccess) return sh_fail(FIR_ERR_COMM, "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

// keys[q] = min over parts p of parts[p][q]   (the shards of one device, before the RCCL call)
__global__ void __launch_bounds__(256) k_shard_min_keys(const uint64_t* __restrict__ parts, int nparts, int n, uint64_t* __restrict__ keys) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t m = kKeyNone;
    for (int p = 0; p < nparts; ++p) {
        const uint64_t v = parts[(size_t)p * n + i];
        m = v < m ? v : m;
    }
    keys[i] = m;
}

// out[q][0..k) = the k smallest of parts[p][q][0..k), p < nparts, ascending. Every part is ascending and keys are unique
// (global row index in the low word), FIR_KEY_NONE sorts last: exactly fir_search_topk over the union of the parts.
// stride: elements from one part to the next (qb * k, or qb * k + 1 when a status element rides behind every part: then
// status_out <- the minimum of those elements).
__global__ void __launch_bounds__(64) k_shard_merge_topk(const uint64_t* __restrict__ parts, int nparts, int qb, int k, uint64_t* __restrict__ out,
                                                          size_t stride, uint64_t* __restrict__ status_out) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (status_out && q == 0) {
        uint64_t m = kKeyNone;
        for (int p = 0; p < nparts; ++p) {
            const uint64_t v = parts[(size_t)p * stride + (size_t)qb * k];
            m = v < m ? v : m;
        }
        *status_out = m;
    }
    if (q >= qb) return;
    uint64_t prev = 0;
    bool first = true;
    for (int j = 0; j < k; ++j) {           // j-th smallest = the smallest key greater than the previous pick
        uint64_t m = kKeyNone;
        for (int p = 0; p < nparts; ++p) {
            const uint64_t* L = parts + (size_t)p * stride + (size_t)q * k;
            for (int i = 0; i < k; ++i) {
                const uint64_t v = L[i];
                if ((first || v > prev) && v < m) m = v;
            }
        }
        out[(size_t)q * k + j] = m;
        if (m == kKeyNone) { for (int r = j + 1; r < k; ++r) out[(size_t)q * k + r] = kKeyNone; break; }
        prev = m;
        first = false;
    }
}

// cls_out[q] = classNo of the row keys[q] names when rows [lo, hi) are held here (cls = their labels), left alone otherwise
__global__ void __launch_bounds__(256) k_shard_class_owned(const uint64_t* __restrict__ keys, int n, const int32_t* __restrict__ cls, int64_t lo,
                                                            int64_t hi, int32_t* __restrict__ cls_out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint64_t v = keys[i];
    if (v == kKeyNone) return;
    const int64_t row = (int64_t)(uint32_t)(v & 0xFFFFFFFFull);
    if (row >= lo && row < hi) cls_out[i] = cls[row - lo];
}
__global__ void __launch_bounds__(256) k_shard_fill_i32(int32_t* p, int n, int32_t v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void __launch_bounds__(256) k_shard_fill_u64(uint64_t* p, int n, uint64_t v) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

struct Worker {       // one thread per device of a multi-device handle: its launches and its RCCL calls
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = false, quit = false;
    int rc = 0;
    char err[512] = "";
    void loop() {
        for (;;) {
            std::function<int()> j;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return has_job || quit; });
                if (quit) return;
                j = job;
                has_job = false;
            }
            const int r = j();
            {
                std::lock_guard<std::mutex> lk(mu);
                rc = r;
                if (r) { strncpy(err, fir_last_error(), sizeof(err) - 1); err[sizeof(err) - 1] = 0; }
                done = true;
            }
            cv.notify_all();
        }
    }
};

struct Shard {
    int slot = 0;              // index into devs
    fir_gallery* g = nullptr;  // nullptr: no rows (more shards than 64-row tiles)
    int64_t lo = 0, hi = 0;    // global rows
    const int32_t* cls = nullptr;
};

struct DevCtx {
    int device = 0;
    hipStream_t stream = nullptr;
    ncclComm_t comm = nullptr;
    float* dq = nullptr;        size_t dq_cap = 0;
    uint64_t* parts = nullptr;  size_t parts_cap = 0;     // [shards here][qb * k]
    uint64_t* keys = nullptr;   size_t keys_cap = 0;      // [qb * k] reduced
    uint64_t* gath = nullptr;   size_t gath_cap = 0;      // [nranks][qb * k]
    int32_t* cls = nullptr;     size_t cls_cap = 0;
    std::vector<int> shards;
    Worker* worker = nullptr;
    std::vector<hipEvent_t> evs;                          // pairs around the RCCL calls (profiling, device slot 0)
    size_t ev_used = 0;
    int32_t* status = nullptr;                            // device: [0] the agreed-growth word, [1] sticky status of the asynchronous calls
    int32_t* h_status = nullptr;                          // pinned host: [0] what this rank contributes, [1] what came back, [2] sticky read-back,
                                                          // [3] the sticky status as the device mirrors it (k_shard_note_status), [4..5] a double (fir_cls_sharded)
    int32_t* d_hsticky = nullptr;                         // device address of h_status[3]
    hipEvent_t async_done = nullptr;                      // behind the last asynchronous device-pointer call, on the stream it ran on
    bool async_pending = false;                           // ... whose status nobody has looked at yet
};

// What the ranks of a handle have to agree on to stay out of each other's way (see "Failure semantics" above).
struct Health {
    std::atomic<bool> dead{false};   // a collective failed or a peer reported an error: no further collective is attempted
    std::atomic<int> dead_code{0};
    int timeout_ms = 120000;         // bound of every wait behind a collective
    int fail_shard = 0;              // test hook: 1-based local shard whose step fails, 0 = none
    int fail_step = 0;               // 1 = its scan (after the agreed growth), 2 = its device's buffer growth
    void apply(const fir_shard_opts& o) {
        if (o.timeout_ms > 0) timeout_ms = o.timeout_ms;
        fail_shard = o.fail_shard;
        fail_step = o.fail_step;
    }
};

template <typename T>
int grow_dev(T*& p, size_t& cap, size_t need) {
    if (need <= cap) return FIR_OK;
    if (p) SH_HIP(hipFree(p));
    p = nullptr;
    cap = 0;
    const size_t want = std::max<size_t>(need, 1024);
    SH_HIP(hipMalloc((void**)&p, want * sizeof(T)));
    cap = want;
    return FIR_OK;
}

int peer_failed(int code) {
    const char* what = code == FIR_ERR_NOMEM ? "allocation failed" : code == FIR_ERR_HIP ? "a HIP call failed" : code == FIR_ERR_COMM ? "an RCCL call failed"
                                                                                                                                    : "its step failed";
    return sh_fail(code < 0 ? code : FIR_ERR_COMM, "a rank of the sharded handle reported an error (%s): the call fails on every rank and the handle is closed for further calls", what);
}

// Abort this device's communicator (a blocked collective returns) and mark the handle dead.
int comm_dead(DevCtx& dc, Health& hl, int code, const char* why) {
    hl.dead = true;
    hl.dead_code = code;
    if (dc.comm) { (void)ncclCommAbort(dc.comm); dc.comm = nullptr; }
    return sh_fail(code, "sharded handle: %s; the communicator was aborted and the handle is closed for further calls", why);
}

// hipStreamSynchronize with a bound: polls the stream and the communicator's asynchronous error state.
int wait_stream(DevCtx& dc, hipStream_t st, Health& hl) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; ++spins) {
        const hipError_t q = hipStreamQuery(st);
        if (q == hipSuccess) return FIR_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return comm_dead(dc, hl, FIR_ERR_HIP, hipGetErrorString(q)); }
        if (dc.comm && (spins & 63) == 63) {
            ncclResult_t ar = ncclSuccess;
            if (ncclCommGetAsyncError(dc.comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                return comm_dead(dc, hl, FIR_ERR_COMM, ncclGetErrorString(ar));
        }
        if (spins > 2000) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > hl.timeout_ms)
                return comm_dead(dc, hl, FIR_ERR_COMM, "a collective did not complete within the time-out (a peer is gone or never entered it)");
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
}

// ... and the same for an event (the asynchronous device-pointer call runs on the CALLER's stream: its completion is an event
// recorded there, not the state of the handle's own stream)
int wait_event(DevCtx& dc, hipEvent_t ev, Health& hl) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int spins = 0;; ++spins) {
        const hipError_t q = hipEventQuery(ev);
        if (q == hipSuccess) return FIR_OK;
        if (q != hipErrorNotReady) { (void)hipGetLastError(); return comm_dead(dc, hl, FIR_ERR_HIP, hipGetErrorString(q)); }
        if (dc.comm && (spins & 63) == 63) {
            ncclResult_t ar = ncclSuccess;
            if (ncclCommGetAsyncError(dc.comm, &ar) == ncclSuccess && ar != ncclSuccess && ar != ncclInProgress)
                return comm_dead(dc, hl, FIR_ERR_COMM, ncclGetErrorString(ar));
        }
        if (spins > 2000) {
            if (std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count() > hl.timeout_ms)
                return comm_dead(dc, hl, FIR_ERR_COMM, "an asynchronous call's exchange did not complete within the time-out (a peer is gone or never entered it)");
            std::this_thread::sleep_for(std::chrono::microseconds(20));
        }
    }
}

// Every rank contributes its local status (FIR_OK or a negative FIR_ERR_*); all of them get the minimum.
int agree(DevCtx& dc, hipStream_t st, Health& hl, int local_rc, int* agreed) {
    if (!dc.comm) return sh_fail(FIR_ERR_STATE, "sharded handle: no communicator");
    dc.h_status[0] = local_rc;
    hipError_t e = hipMemcpyAsync(dc.status, dc.h_status, sizeof(int32_t), hipMemcpyHostToDevice, st);
    ncclResult_t nr = e == hipSuccess ? ncclAllReduce(dc.status, dc.status, 1, ncclInt32, ncclMin, dc.comm, st) : ncclSuccess;
    if (e == hipSuccess && nr == ncclSuccess) e = hipMemcpyAsync(dc.h_status + 1, dc.status, sizeof(int32_t), hipMemcpyDeviceToHost, st);
    if (e != hipSuccess) return comm_dead(dc, hl, FIR_ERR_HIP, hipGetErrorString(e));
    if (nr != ncclSuccess) return comm_dead(dc, hl, FIR_ERR_COMM, ncclGetErrorString(nr));
    const int rc = wait_stream(dc, st, hl);
    if (rc) return rc;
    *agreed = dc.h_status[1];
    return FIR_OK;
}

// The agreed growth step of a call: `need` says whether this call grows anything on this rank (the same answer on every
// rank), `alloc` tries the allocations. Everyone leaves with the same verdict.
int grow_agreed(DevCtx& dc, hipStream_t st, Health& hl, bool need, bool inject, const std::function<int()>& alloc) {
    if (!need) return FIR_OK;
    int r = inject ? sh_fail(FIR_ERR_NOMEM, "injected allocation failure (fir_shard_opts.fail_step = 2)") : alloc();
    char mine[512];
    if (r) { strncpy(mine, fir_last_error(), sizeof(mine) - 1); mine[sizeof(mine) - 1] = 0; }
    int all = FIR_OK;
    const int ra = agree(dc, st, hl, r, &all);
    if (ra) return ra;
    if (all == FIR_OK) return FIR_OK;
    hl.dead = true;
    hl.dead_code = all;
    if (r) { fir_set_last_error_(mine); return r; }       // this rank's own error text
    return peer_failed(all);
}

constexpr uint64_t kStatusOk = kKeyNone;                  // the element behind a key payload: all ones = every rank was fine
__host__ __device__ inline uint64_t status_key(int rc) { return rc == FIR_OK ? kStatusOk : (uint64_t)(uint32_t)(1000 + rc); }   // smaller = worse under ncclMin
inline int status_code(uint64_t v) { return v == kStatusOk ? FIR_OK : (int)(int64_t)v - 1000; }

__global__ void k_shard_set_u64(uint64_t* p, uint64_t v) { *p = v; }
__global__ void k_shard_set_i32(int32_t* p, int32_t v) { *p = v; }
__global__ void k_shard_set_f64(double* p, double v) { *p = v; }
// sticky[0] = min(sticky[0], status of the exchange just finished) -- the asynchronous device-pointer call's record
// ... mirrored into pinned host memory, where the next call (or fir_sharded_sync) reads it without a copy of its own
__global__ void k_shard_note_status(const uint64_t* __restrict__ st, int32_t* __restrict__ sticky, int32_t* __restrict__ host_sticky) {
    const uint64_t v = *st;
    const int32_t c = v == kKeyNone ? 0 : (int32_t)(int64_t)v - 1000;
    if (c < *sticky) { *sticky = c; *host_sticky = c; }
}

}  // namespace

struct fir_sharded {
    int exchanges_done = 0;          // (audit build's fail_step = 3)
    int d = 0, metric = 0;
    int64_t n_local = 0, first_row = 0;
    int ndev = 0, spd = 1, nranks = 1, rank0 = 0;
    bool has_labels = false;
    std::vector<DevCtx> devs;
    std::vector<Shard> shards;
    void* pin = nullptr;  size_t pin_cap = 0;      // pinned host staging: queries in, keys / classes out
    bool profiling = false;
    Health health;
};

namespace {

int setup_devices(std::vector<DevCtx>& devs, const int32_t* devices, int ndev, int nranks, int first_rank, const void* comm_id);
void teardown_devices(std::vector<DevCtx>& devs, bool dead = false);

// Run fn(slot) for every device: inline for one device, on the per-device worker threads otherwise.
int run_all(std::vector<DevCtx>& devs, const std::function<int(int)>& fn);
int run_all(fir_sharded* h, const std::function<int(int)>& fn) { return run_all(h->devs, fn); }
int run_all(std::vector<DevCtx>& devs, const std::function<int(int)>& fn) {
    const int ndev = (int)devs.size();
    struct { std::vector<DevCtx>& devs; int ndev; } hh{devs, ndev};
    auto* h = &hh;
    if (h->ndev == 1) return fn(0);
    for (int s = 0; s < h->ndev; ++s) {
        Worker* w = h->devs[s].worker;
        {
            std::lock_guard<std::mutex> lk(w->mu);
            w->job = [fn, s] { return fn(s); };
            w->has_job = true;
            w->done = false;
        }
        w->cv.notify_all();
    }
    int rc = FIR_OK;
    for (int s = 0; s < h->ndev; ++s) {
        Worker* w = h->devs[s].worker;
        std::unique_lock<std::mutex> lk(w->mu);
        w->cv.wait(lk, [&] { return w->done; });
        if (w->rc && rc == FIR_OK) { rc = w->rc; fir_set_last_error_(w->err); }
    }
    return rc;
}

int ensure_pin(fir_sharded* h, size_t bytes) {
    if (bytes <= h->pin_cap) return FIR_OK;
    if (h->pin) SH_HIP(hipHostFree(h->pin));
    h->pin = nullptr;
    h->pin_cap = 0;
    const size_t want = std::max<size_t>(bytes, 1 << 20);
    SH_HIP(hipHostMalloc(&h->pin, want, hipHostMallocPortable));
    h->pin_cap = want;
    return FIR_OK;
}

int check_range(const fir_sharded* h, int32_t& start, int32_t& end) {
    if (end == 0) end = h->d;
    if (start < 0 || end > h->d || start >= end) return sh_fail(FIR_ERR_ARG, "feature range [%d,%d) not inside [0,%d)", start, end, h->d);
    return FIR_OK;
}

// The buffers of one call on one device, grown in the call's agreed step: queries (host-pointer form), the shards' partial
// keys, the reduced keys + status element, the gathered keys of all ranks (top-K) and the winners' classes + status element.
struct CallBuffers { size_t dq, parts, keys, gath, cls; };
CallBuffers call_buffers(const fir_sharded* h, const DevCtx& dc, int32_t qb, int32_t k, bool host_queries, bool classes) {
    const size_t per = (size_t)qb * k;
    return {host_queries ? (size_t)qb * h->d : 0, per * std::max<size_t>(dc.shards.size(), 1), per + 1,
            k > 1 ? (per + 1) * (size_t)h->nranks : 0, classes ? (size_t)qb + 1 : 0};
}
bool call_grows(const DevCtx& dc, const CallBuffers& b) {
    return b.dq > dc.dq_cap || b.parts > dc.parts_cap || b.keys > dc.keys_cap || b.gath > dc.gath_cap || b.cls > dc.cls_cap;
}
int call_alloc(DevCtx& dc, const CallBuffers& b) {
    int rc;
    if ((rc = grow_dev(dc.dq, dc.dq_cap, b.dq))) return rc;
    if ((rc = grow_dev(dc.parts, dc.parts_cap, b.parts))) return rc;
    if ((rc = grow_dev(dc.keys, dc.keys_cap, b.keys))) return rc;
    if ((rc = grow_dev(dc.gath, dc.gath_cap, b.gath))) return rc;
    return grow_dev(dc.cls, dc.cls_cap, b.cls);
}
bool inject_here(const fir_sharded* h, const DevCtx& dc, int step) {
    if (h->health.fail_step != step || h->health.fail_shard <= 0) return false;
    for (int si : dc.shards)
        if (si == h->health.fail_shard - 1) return true;
    return false;
}

// The keys of one device: every shard it holds scans the batch (K keys per query, K = 1: top-1), then the minimum /
// K-way merge over those shards lands in dc.keys[0, qb * k). Buffers exist (call_alloc).
int device_keys(fir_sharded* h, int slot, const float* d_queries, int32_t qb, int32_t start, int32_t end, int32_t k, hipStream_t st) {
    DevCtx& dc = h->devs[slot];
    const size_t per = (size_t)qb * k;
    int rc;
    int live = 0;
    for (size_t i = 0; i < dc.shards.size(); ++i) {
        Shard& s = h->shards[dc.shards[i]];
        if (h->health.fail_step == 1 && h->health.fail_shard - 1 == dc.shards[i])
            return sh_fail(FIR_ERR_NOMEM, "injected scan failure on shard %d (fir_shard_opts.fail_step = 1)", dc.shards[i]);
        if (!s.g) continue;
        uint64_t* out = dc.parts + (size_t)live * per;
        rc = k == 1 ? fir_search_top1_keys_dev(s.g, d_queries, qb, start, end, out, st)
                    : fir_search_topk_keys_dev(s.g, d_queries, qb, start, end, k, out, st);
        if (rc) return rc;
        ++live;
    }
    if (live == 0) hipLaunchKernelGGL(k_shard_fill_u64, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, dc.keys, (int)per, kKeyNone);
    else if (k == 1) hipLaunchKernelGGL(k_shard_min_keys, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, dc.parts, live, (int)per, dc.keys);
    else hipLaunchKernelGGL(k_shard_merge_topk, dim3((qb + 63) / 64), dim3(64), 0, st, dc.parts, live, qb, k, dc.keys, per, (uint64_t*)nullptr);
    SH_HIP(hipGetLastError());
    return FIR_OK;
}

// The exchange step over all ranks, on `st`, result in dc.keys (in place); dc.keys[qb * k] carries local_rc in and the worst
// status of all ranks out. A rank whose scans failed contributes FIR_KEY_NONE keys. Errors here kill the communicator.
int exchange_keys(fir_sharded* h, int slot, int32_t qb, int32_t k, hipStream_t st, int local_rc) {
    DevCtx& dc = h->devs[slot];
    uint64_t* keys = dc.keys;
    const size_t per = (size_t)qb * k;
    const bool prof = h->profiling && slot == 0;
    if (local_rc) hipLaunchKernelGGL(k_shard_fill_u64, dim3((unsigned)((per + 255) / 256)), dim3(256), 0, st, keys, (int)per, kKeyNone);
    hipLaunchKernelGGL(k_shard_set_u64, dim3(1), dim3(1), 0, st, keys + per, status_key(local_rc));
    if (prof) {
        if (dc.ev_used + 2 > dc.evs.size())
            for (int i = 0; i < 64; ++i) {
                hipEvent_t e;
                if (hipEventCreate(&e) != hipSuccess) break;
                dc.evs.push_back(e);
            }
        if (dc.ev_used + 2 <= dc.evs.size()) (void)hipEventRecord(dc.evs[dc.ev_used], st);
    }
    ncclResult_t nr;
    if (k == 1) {
        nr = ncclAllReduce(keys, keys, per + 1, ncclUint64, ncclMin, dc.comm, st);
    } else {
        nr = ncclAllGather(keys, dc.gath, per + 1, ncclUint64, dc.comm, st);
        if (nr == ncclSuccess)
            hipLaunchKernelGGL(k_shard_merge_topk, dim3((qb + 63) / 64), dim3(64), 0, st, dc.gath, h->nranks, qb, k, keys, per + 1, keys + per);
    }
    if (nr != ncclSuccess) return comm_dead(dc, h->health, FIR_ERR_COMM, ncclGetErrorString(nr));
#ifdef FIR_AUDIT
    // fail_step = 3 (audit build): from the handle's SECOND exchange on the status element comes BACK poisoned, as if a peer's scan had
    // failed -- this rank's own work is fine (the first call of a handle grows its buffers in the agreed, blocking step: a test of the
    // asynchronous path needs one clean call in front)
    if (inject_here(h, dc, 3) && h->exchanges_done++ >= 1) hipLaunchKernelGGL(k_shard_set_u64, dim3(1), dim3(1), 0, st, keys + per, status_key(FIR_ERR_NOMEM));
#endif
    if (prof && dc.ev_used + 2 <= dc.evs.size()) {
        (void)hipEventRecord(dc.evs[dc.ev_used + 1], st);
        dc.ev_used += 2;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return comm_dead(dc, h->health, FIR_ERR_HIP, hipGetErrorString(e));
    return FIR_OK;
}

// class of every winning row: the rank that holds the row knows it, everyone else contributes INT32_MAX; dc.cls[qb] is the
// status element (INT32_MAX = fine, a negative FIR_ERR_* otherwise: the minimum finds it)
int device_classes(fir_sharded* h, int slot, const uint64_t* keys, int32_t qb, hipStream_t st, int local_rc) {
    DevCtx& dc = h->devs[slot];
    hipLaunchKernelGGL(k_shard_fill_i32, dim3((qb + 255) / 256), dim3(256), 0, st, dc.cls, qb, 0x7FFFFFFF);
    hipLaunchKernelGGL(k_shard_set_i32, dim3(1), dim3(1), 0, st, dc.cls + qb, local_rc ? local_rc : 0x7FFFFFFF);
    if (!local_rc)
        for (int si : dc.shards) {
            const Shard& s = h->shards[si];
            if (!s.g || !s.cls) continue;
            hipLaunchKernelGGL(k_shard_class_owned, dim3((qb + 255) / 256), dim3(256), 0, st, keys, qb, s.cls, s.lo, s.hi, dc.cls);
        }
    const ncclResult_t nr = ncclAllReduce(dc.cls, dc.cls, (size_t)qb + 1, ncclInt32, ncclMin, dc.comm, st);
    if (nr != ncclSuccess) return comm_dead(dc, h->health, FIR_ERR_COMM, ncclGetErrorString(nr));
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return comm_dead(dc, h->health, FIR_ERR_HIP, hipGetErrorString(e));
    return FIR_OK;
}

int dead_handle(const Health& hl) {
    return sh_fail(FIR_ERR_STATE, "this sharded handle is closed: an earlier call failed on some rank (code %d); destroy it and create a fresh one", hl.dead_code.load());
}

// Host-pointer search: queries go to every device from pinned memory, keys (and classes) come back from device slot 0.
int search_host(fir_sharded* h, const float* queries, int32_t qb, int32_t start, int32_t end, int32_t k, int32_t* idx, float* dist,
                int32_t* class_out) {
    if (!h || (qb > 0 && !queries)) return sh_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return sh_fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > 8) return sh_fail(FIR_ERR_ARG, "k=%d outside [1,8]", k);
    if (class_out && !h->has_labels) return sh_fail(FIR_ERR_STATE, "the sharded gallery was created without class labels");
    if (h->health.dead) return dead_handle(h->health);
    if (qb == 0) return FIR_OK;
    int rc = check_range(h, start, end);
    if (rc) return rc;
    // (argument errors above are the same on every rank: nobody has entered a collective yet)
    const size_t per = (size_t)qb * k;
    const size_t qbytes = (size_t)qb * h->d * sizeof(float), kbytes = (per + 1) * sizeof(uint64_t), cbytes = ((size_t)qb + 1) * sizeof(int32_t);
    // the pinned staging area is part of the agreed growth as well: slot 0 tries it inside the step
    float* hq = nullptr;
    uint64_t* hk = nullptr;
    int32_t* hc = nullptr;
    std::mutex pin_mu;
    std::condition_variable pin_cv;
    bool pin_ready = false;
    int pin_rc = FIR_OK;
    const bool pin_grows = qbytes + kbytes + cbytes > h->pin_cap;            // read before slot 0 changes it: the same answer for every slot
    rc = run_all(h, [&](int slot) -> int {
        DevCtx& dc = h->devs[slot];
        SH_HIP(hipSetDevice(dc.device));
        const CallBuffers nb = call_buffers(h, dc, qb, k, true, class_out != nullptr);
        int r = grow_agreed(dc, dc.stream, h->health, call_grows(dc, nb) || pin_grows, inject_here(h, dc, 2), [&]() -> int {
            int ra = call_alloc(dc, nb);
            if (slot == 0 && !ra) ra = ensure_pin(h, qbytes + kbytes + cbytes);
            return ra;
        });
        if (slot == 0) {
            // the other devices of this process read the queries from the staging area slot 0 has just made sure of
            if (!r) {
                hq = (float*)h->pin;
                hk = (uint64_t*)((char*)h->pin + qbytes);
                hc = (int32_t*)((char*)h->pin + qbytes + kbytes);
                std::memcpy(hq, queries, qbytes);
            }
            { std::lock_guard<std::mutex> lk(pin_mu); pin_ready = true; pin_rc = r; }
            pin_cv.notify_all();
        } else {
            std::unique_lock<std::mutex> lk(pin_mu);
            pin_cv.wait(lk, [&] { return pin_ready; });
            if (!r && pin_rc) r = pin_rc;          // (an agreed failure has r != 0 everywhere already)
        }
        if (r) return r;
        // from here on every rank takes part in every collective of the call, whatever happens to it
        int local = FIR_OK;
        char mine[512] = "";
        if (hipMemcpyAsync(dc.dq, hq, qbytes, hipMemcpyHostToDevice, dc.stream) != hipSuccess) local = sh_fail(FIR_ERR_HIP, "query upload failed");
        if (!local) local = device_keys(h, slot, dc.dq, qb, start, end, k, dc.stream);
        if (local) { strncpy(mine, fir_last_error(), sizeof(mine) - 1); (void)hipGetLastError(); }
        if ((r = exchange_keys(h, slot, qb, k, dc.stream, local))) return r;
        if (class_out && (r = device_classes(h, slot, dc.keys, qb, dc.stream, local))) return r;
        if (slot == 0) {
            (void)hipMemcpyAsync(hk, dc.keys, kbytes, hipMemcpyDeviceToHost, dc.stream);
            if (class_out) (void)hipMemcpyAsync(hc, dc.cls, cbytes, hipMemcpyDeviceToHost, dc.stream);
        } else {
            (void)hipMemcpyAsync(dc.h_status + 1, dc.keys + per, sizeof(int32_t), hipMemcpyDeviceToHost, dc.stream);   // low word of the status element
        }
        if ((r = wait_stream(dc, dc.stream, h->health))) return r;
        if (local) { h->health.dead = true; h->health.dead_code = local; fir_set_last_error_(mine); return local; }
        return FIR_OK;
    });
    if (rc) { h->health.dead = true; if (!h->health.dead_code.load()) h->health.dead_code = rc; return rc; }
    int all = status_code(hk[per]);
    if (!all && class_out && hc[qb] != 0x7FFFFFFF) all = hc[qb];
    if (all) { h->health.dead = true; h->health.dead_code = all; return peer_failed(all); }
    if (class_out)
        for (int i = 0; i < qb; ++i) class_out[i] = hc[i] == 0x7FFFFFFF ? -1 : hc[i];     // ImageTesting.cpp:63: no row found -> -1
    if (idx || dist) return fir_keys_unpack(hk, qb * k, idx, dist);
    return FIR_OK;
}

}  // namespace

namespace {

// per-device stream, communicator (rank = first_rank + slot) and, for several devices, worker threads
int setup_devices(std::vector<DevCtx>& devs, const int32_t* devices, int ndev, int nranks, int first_rank, const void* comm_id) {
    ncclUniqueId id;
    if (comm_id) std::memcpy(&id, comm_id, sizeof id);
    else SH_NCCL(ncclGetUniqueId(&id));
    for (int s = 0; s < ndev; ++s) {
        DevCtx& dc = devs[(size_t)s];
        dc.device = devices[s];
        SH_HIP(hipSetDevice(dc.device));
        SH_HIP(hipStreamCreateWithFlags(&dc.stream, hipStreamNonBlocking));
    }
    for (int s = 0; s < ndev; ++s) {
        DevCtx& dc = devs[(size_t)s];
        SH_HIP(hipSetDevice(dc.device));
        SH_HIP(hipMalloc((void**)&dc.status, 4 * sizeof(int32_t)));
        SH_HIP(hipMemset(dc.status, 0, 4 * sizeof(int32_t)));
        SH_HIP(hipHostMalloc((void**)&dc.h_status, 8 * sizeof(int32_t), hipHostMallocPortable | hipHostMallocMapped));
        std::memset(dc.h_status, 0, 8 * sizeof(int32_t));
        void* dp = nullptr;
        SH_HIP(hipHostGetDevicePointer(&dp, dc.h_status, 0));
        dc.d_hsticky = (int32_t*)dp + 3;
        SH_HIP(hipEventCreateWithFlags(&dc.async_done, hipEventDisableTiming));
    }
    // RCCL writes its version banner to STDOUT when a process's FIRST communicator comes up. The callers of this library print
    // results there (the reference's harnesses, whose output must stay diffable): for that one creation -- serialised by a
    // process-wide mutex -- file descriptor 1 points at stderr. Later communicators print nothing and touch no descriptor.
    static std::mutex banner_mu;
    static bool banner_done = false;
    ncclResult_t nr;
    {
        std::unique_lock<std::mutex> lk(banner_mu);
        const bool first = !banner_done;
        int saved_stdout = -1;
        if (first) {
            std::fflush(stdout);
            saved_stdout = dup(1);
            if (saved_stdout >= 0) (void)dup2(2, 1);
        } else {
            lk.unlock();
        }
        nr = ncclGroupStart();
        for (int s = 0; s < ndev && nr == ncclSuccess; ++s) {
            (void)hipSetDevice(devs[(size_t)s].device);
            nr = ncclCommInitRank(&devs[(size_t)s].comm, nranks, id, first_rank + s);
        }
        if (nr == ncclSuccess) nr = ncclGroupEnd(); else (void)ncclGroupEnd();
        if (first) {
            std::fflush(stdout);
            if (saved_stdout >= 0) { (void)dup2(saved_stdout, 1); (void)close(saved_stdout); }
            banner_done = true;
        }
    }
    if (nr != ncclSuccess) return sh_fail(FIR_ERR_COMM, "RCCL communicator of %d ranks: %s", nranks, ncclGetErrorString(nr));
    if (ndev > 1)
        for (int s = 0; s < ndev; ++s) {
            Worker* w = new (std::nothrow) Worker();
            if (!w) return sh_fail(FIR_ERR_NOMEM, "host allocation failed");
            devs[(size_t)s].worker = w;
            w->th = std::thread([w] { w->loop(); });
        }
    return FIR_OK;
}

void teardown_devices(std::vector<DevCtx>& devs, bool dead) {
    for (DevCtx& dc : devs) {
        if (dc.worker) {
            { std::lock_guard<std::mutex> lk(dc.worker->mu); dc.worker->quit = true; }
            dc.worker->cv.notify_all();
            if (dc.worker->th.joinable()) dc.worker->th.join();
            delete dc.worker;
            dc.worker = nullptr;
        }
        (void)hipSetDevice(dc.device);
        if (dc.stream && !dead) (void)hipStreamSynchronize(dc.stream);
        if (dc.comm) { if (dead) (void)ncclCommAbort(dc.comm); else (void)ncclCommDestroy(dc.comm); }
        (void)hipFree(dc.status);
        if (dc.h_status) (void)hipHostFree(dc.h_status);
        if (dc.async_done) (void)hipEventDestroy(dc.async_done);
        (void)hipFree(dc.dq); (void)hipFree(dc.parts); (void)hipFree(dc.keys); (void)hipFree(dc.gath); (void)hipFree(dc.cls);
        for (hipEvent_t e : dc.evs) (void)hipEventDestroy(e);
        if (dc.stream) (void)hipStreamDestroy(dc.stream);
        dc = DevCtx();
    }
}

int check_devices(const int32_t* devices, int32_t ndev) {
    if (!devices || ndev < 1 || ndev > 64) return sh_fail(FIR_ERR_ARG, "device list of %d entries", ndev);
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return sh_fail(FIR_ERR_ARG, "device %d listed twice (use shards_per_device for logical shards)", devices[i]);
    const int cnt = fir_device_count();
    for (int i = 0; i < ndev; ++i)
        if (devices[i] < 0 || devices[i] >= cnt) return sh_fail(FIR_ERR_NODEVICE, "device %d out of range (%d visible)", devices[i], cnt);
    return FIR_OK;
}

int read_opts(const fir_shard_opts* opts, fir_shard_opts* o) {
    std::memset(o, 0, sizeof *o);
    if (opts) {
        if (opts->struct_bytes < 8 || opts->struct_bytes > (int32_t)sizeof *o) return sh_fail(FIR_ERR_ARG, "fir_shard_opts.struct_bytes = %d", opts->struct_bytes);
        std::memcpy(o, opts, (size_t)opts->struct_bytes);
    }
    if (o->shards_per_device <= 0) o->shards_per_device = 1;
    if (!o->comm_id) { o->nprocs = 1; o->proc_rank = 0; }
    if (o->shards_per_device > 64) return sh_fail(FIR_ERR_ARG, "shards_per_device = %d", o->shards_per_device);
    if (o->nprocs < 1 || o->proc_rank < 0 || o->proc_rank >= o->nprocs) return sh_fail(FIR_ERR_ARG, "process %d of %d", o->proc_rank, o->nprocs);
#ifndef FIR_AUDIT
    // the failure-injection hooks exist in the audit build only (libfir_amd_audit.so): the shipped library refuses them loudly
    if (o->fail_shard != 0 || o->fail_step != 0)
        return sh_fail(FIR_ERR_ARG, "fir_shard_opts.fail_shard / fail_step are test hooks of the audit build (libfir_amd_audit.so); this library has none");
#endif
    return FIR_OK;
}

// acc[i] += part[i]
__global__ void __launch_bounds__(256) k_shard_add_f64(double* __restrict__ acc, const double* __restrict__ part, int64_t n, int first) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) acc[i] = first ? part[i] : acc[i] + part[i];
}
// best[q] = first maximum of scores[q][0..C) from -DBL_MAX (classification.cpp:217-224); one wave per query
__global__ void __launch_bounds__(64) k_shard_argmax_first(const double* __restrict__ scores, int num_classes, int32_t* __restrict__ best) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const double* s = scores + (size_t)q * num_classes;
    double mx = -1.7976931348623157e308;
    int bi = 0x7FFFFFFF;
    for (int i = lane; i < num_classes; i += 64)
        if (mx < s[i]) { mx = s[i]; bi = i; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const double om = __shfl_xor(mx, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (om > mx || (om == mx && oi < bi)) { mx = om; bi = oi; }
    }
    if (lane == 0) best[q] = bi == 0x7FFFFFFF ? -1 : bi;
}

// merged[q][c][0..k) = the k smallest of parts[p][q][c][0..k), p < nparts (every part ascending, DBL_MAX = no such row)
// stride: doubles from one part to the next (qc * k, or qc * k + 1 with a status element behind every part: then
// status_out <- the largest of those elements; 0 = every rank was fine)
__global__ void __launch_bounds__(256) k_shard_merge_knn(const double* __restrict__ parts, int nparts, int64_t qc /* qb * C */, int k, double* __restrict__ merged,
                                                          size_t stride, double* __restrict__ status_out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (status_out && i == 0) {
        double m = 0.0;
        for (int p = 0; p < nparts; ++p) m = fmax(m, parts[(size_t)p * stride + (size_t)qc * k]);
        *status_out = m;
    }
    if (i >= qc) return;
    double best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) best[j] = 1.7976931348623157e308;
    for (int p = 0; p < nparts; ++p)
        for (int j = 0; j < k; ++j) {
            double v = parts[(size_t)p * stride + (size_t)i * k + j];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const bool sw = v < best[t];
                const double tmp = best[t];
                best[t] = sw ? v : tmp;
                v = sw ? tmp : v;
            }
        }
    for (int j = 0; j < k; ++j) merged[(size_t)i * k + j] = best[j];
}
// KNNClassifier::predict's vote (classification.cpp:151-168) from the merged lists: the class that first collects k votes in the
// globally sorted order is the one whose k-th nearest member is nearest (first minimum); when no class has k rows the
// reference's loop ends without a break and the largest class (first on ties) wins. One thread per query.
__global__ void __launch_bounds__(64) k_shard_knn_vote(const double* __restrict__ merged, int qb, int num_classes, int k,
                                                       const int32_t* __restrict__ class_count, int32_t* __restrict__ best) {
    const int q = blockIdx.x * 64 + threadIdx.x;
    if (q >= qb) return;
    double mn = 1.7976931348623157e308;
    int b = -1;
    for (int c = 0; c < num_classes; ++c) {
        const double v = class_count[c] >= k ? merged[((size_t)q * num_classes + c) * k + (k - 1)] : 1.7976931348623157e308;
        if (v < mn) { mn = v; b = c; }
    }
    if (b < 0) {
        int mc = -1;
        for (int c = 0; c < num_classes; ++c)
            if (class_count[c] > mc) { mc = class_count[c]; b = c; }
    }
    best[q] = b;
}

}  // namespace

struct fir_cls_sharded {
    int d = 0, num_classes = 0, ndev = 0, nranks = 1;
    int64_t nt_local = 0;
    std::vector<DevCtx> devs;
    struct Part { int slot; fir_cls* c; };
    std::vector<Part> parts;
    std::vector<double*> acc;      size_t acc_cap = 0;     // per device: summed class scores [qb][C]
    std::vector<int32_t*> best;                            // per device
    std::vector<int32_t*> class_count;                     // per device: training rows per class over ALL shards and ranks
    std::vector<double*> knn;      size_t knn_cap = 0;     // per device: [shards here + nranks][qb][C][k] lists being merged
    size_t best_cap = 0;                                   // per device: ints in best[]
    void* pin = nullptr;  size_t pin_cap = 0;
    Health health;
};

extern "C" {

int fir_cls_sharded_destroy(fir_cls_sharded* h) {
    if (!h) return FIR_OK;
    for (auto& p : h->parts)
        if (p.c) fir_cls_destroy(p.c);
    for (size_t s = 0; s < h->devs.size(); ++s) {
        (void)hipSetDevice(h->devs[s].device);
        if (s < h->acc.size()) (void)hipFree(h->acc[s]);
        if (s < h->best.size()) (void)hipFree(h->best[s]);
        if (s < h->class_count.size()) (void)hipFree(h->class_count[s]);
        if (s < h->knn.size()) (void)hipFree(h->knn[s]);
    }
    teardown_devices(h->devs, h->health.dead);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return FIR_OK;
}

int fir_cls_create_sharded(const double* train_rows, int64_t nt, int32_t d, const int32_t* train_class, int32_t num_classes, const double* avg,
                           const int32_t* devices, int32_t ndev, const fir_shard_opts* opts, fir_cls_sharded** out) {
    if (!out) return sh_fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (nt < 0 || d <= 0 || num_classes <= 0 || !avg || (nt > 0 && (!train_rows || !train_class))) return sh_fail(FIR_ERR_ARG, "bad training set (nt=%lld d=%d classes=%d)", (long long)nt, d, num_classes);
    int rc = check_devices(devices, ndev);
    if (rc) return rc;
    fir_shard_opts o;
    if ((rc = read_opts(opts, &o))) return rc;
    if (o.rows_on_device) return sh_fail(FIR_ERR_ARG, "fir_cls_create_sharded takes host rows");
    fir_cls_sharded* h = new (std::nothrow) fir_cls_sharded();
    if (!h) return sh_fail(FIR_ERR_NOMEM, "host allocation failed");
    h->d = d; h->num_classes = num_classes; h->ndev = ndev; h->nranks = o.nprocs * ndev; h->nt_local = nt;
    h->health.apply(o);
    h->devs.resize((size_t)ndev);
    h->acc.assign((size_t)ndev, nullptr);
    h->best.assign((size_t)ndev, nullptr);
    h->class_count.assign((size_t)ndev, nullptr);
    h->knn.assign((size_t)ndev, nullptr);
    const int spd = o.shards_per_device, nsh = ndev * spd;
    const int64_t total = o.total_rows > 0 ? o.total_rows : nt;
    const int64_t per = (((nt + 63) / 64 + nsh - 1) / nsh) * 64;
    for (int s = 0; s < nsh && rc == FIR_OK; ++s) {
        const int64_t lo = std::min<int64_t>((int64_t)s * per, nt), hi = std::min<int64_t>((int64_t)(s + 1) * per, nt);
        fir_cls_sharded::Part p{s / spd, nullptr};
        if (hi > lo) {
            rc = fir_cls_create(train_rows + lo * d, hi - lo, d, train_class + lo, num_classes, avg, devices[p.slot], &p.c);
            if (rc == FIR_OK) rc = fir_cls_set_total_training_size(p.c, total);     // classification.cpp:215: every partial sum over the global N
        }
        h->parts.push_back(p);
    }
    if (rc == FIR_OK) rc = setup_devices(h->devs, devices, ndev, h->nranks, o.proc_rank * ndev, o.comm_id);
    // rows per class over all shards and ranks (the kNN vote's "class has k rows" and its largest-class rule)
    if (rc == FIR_OK) {
        std::vector<int32_t> cnt((size_t)num_classes, 0);
        for (int64_t t = 0; t < nt; ++t)
            if (train_class[t] >= 0 && train_class[t] < num_classes) cnt[(size_t)train_class[t]]++;
        rc = run_all(h->devs, [&](int slot) -> int {
            DevCtx& dc = h->devs[(size_t)slot];
            SH_HIP(hipSetDevice(dc.device));
            SH_HIP(hipMalloc((void**)&h->class_count[(size_t)slot], (size_t)num_classes * sizeof(int32_t)));
            int32_t* cc = h->class_count[(size_t)slot];
            if (slot == 0) SH_HIP(hipMemcpyAsync(cc, cnt.data(), (size_t)num_classes * sizeof(int32_t), hipMemcpyHostToDevice, dc.stream));
            else SH_HIP(hipMemsetAsync(cc, 0, (size_t)num_classes * sizeof(int32_t), dc.stream));      // this process's rows are counted once
            SH_NCCL(ncclAllReduce(cc, cc, (size_t)num_classes, ncclInt32, ncclSum, dc.comm, dc.stream));
            return wait_stream(dc, dc.stream, h->health);
        });
    }
    if (rc) { fir_cls_sharded_destroy(h); return rc; }
    *out = h;
    return FIR_OK;
}

// a part of this device whose step is made to fail (test hook), by its index among all parts of the handle
static bool cls_inject(const fir_cls_sharded* h, int part_index, int step) { return h->health.fail_step == step && h->health.fail_shard - 1 == part_index; }
static bool cls_inject_dev(const fir_cls_sharded* h, int slot, int step) {
    for (size_t i = 0; i < h->parts.size(); ++i)
        if (h->parts[i].slot == slot && cls_inject(h, (int)i, step)) return true;
    return false;
}

int fir_cls_sharded_knn_predict(fir_cls_sharded* h, const double* queries, int32_t qb, int32_t k, int32_t* best_class) {
    if (!h || !best_class || (qb > 0 && !queries)) return sh_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return sh_fail(FIR_ERR_ARG, "qb < 0");
    if (k < 1 || k > 8) return sh_fail(FIR_ERR_ARG, "k=%d outside [1,8]", k);
    if (h->health.dead) return dead_handle(h->health);
    if (qb == 0) return FIR_OK;
    int32_t maxb = 1 << 30;
    for (auto& p : h->parts)
        if (p.c) { int32_t mb = 0; double* ds; void* st; fir_cls_knn_nearest_dev_(p.c, nullptr, 0, k, &ds, &st, &mb); maxb = std::min(maxb, mb); }
    maxb = std::min(maxb, 1024);
    if (qb > maxb) {
        for (int32_t q0 = 0; q0 < qb; q0 += maxb) {
            const int rc0 = fir_cls_sharded_knn_predict(h, queries + (size_t)q0 * h->d, std::min(maxb, qb - q0), k, best_class + q0);
            if (rc0) return rc0;
        }
        return FIR_OK;
    }
    const size_t per = (size_t)qb * h->num_classes * k;                       // doubles in one [qb][C][k] table
    size_t max_parts = 1;
    for (int s = 0; s < h->ndev; ++s) {
        size_t np = 0;
        for (auto& p : h->parts) np += (p.slot == s && p.c) ? 1 : 0;
        max_parts = std::max(max_parts, np);
    }
    // [max_parts] shard tables | their merge + status element | [nranks] gathered tables + status elements | the merge over the ranks + status
    const size_t need = per * max_parts + (per + 1) + (per + 1) * (size_t)h->nranks + (per + 1);
    const size_t pin_need = ((size_t)qb + 2) * sizeof(int32_t) + sizeof(double);
    const bool grows = need > h->knn_cap || (size_t)qb > h->best_cap || pin_need > h->pin_cap;     // the same answer on every rank
    const size_t new_best = std::max<size_t>(4096, (size_t)qb);
    const int64_t qc = (int64_t)qb * h->num_classes;
    int rc = run_all(h->devs, [&](int slot) -> int {
        DevCtx& dc = h->devs[(size_t)slot];
        SH_HIP(hipSetDevice(dc.device));
        int r = grow_agreed(dc, dc.stream, h->health, grows, cls_inject_dev(h, slot, 2), [&]() -> int {
            if (need > h->knn_cap) {
                if (h->knn[(size_t)slot]) SH_HIP(hipFree(h->knn[(size_t)slot]));
                h->knn[(size_t)slot] = nullptr;
                SH_HIP(hipMalloc((void**)&h->knn[(size_t)slot], need * sizeof(double)));
            }
            if ((size_t)qb > h->best_cap) {
                if (h->best[(size_t)slot]) SH_HIP(hipFree(h->best[(size_t)slot]));
                h->best[(size_t)slot] = nullptr;
                SH_HIP(hipMalloc((void**)&h->best[(size_t)slot], new_best * sizeof(int32_t)));
            }
            if (slot == 0 && pin_need > h->pin_cap) {
                if (h->pin) SH_HIP(hipHostFree(h->pin));
                h->pin = nullptr; h->pin_cap = 0;
                SH_HIP(hipHostMalloc(&h->pin, std::max<size_t>(pin_need, 4096 * sizeof(int32_t) + 64), hipHostMallocPortable));
                h->pin_cap = std::max<size_t>(pin_need, 4096 * sizeof(int32_t) + 64);
            }
            return FIR_OK;
        });
        if (r) return r;
        double* parts = h->knn[(size_t)slot];                                 // [max_parts] tables of the shards held here
        double* mine = parts + per * max_parts;                               // their merge (+ status element)
        double* gath = mine + per + 1;                                        // [nranks] tables (+ status elements)
        double* all = gath + (per + 1) * (size_t)h->nranks;                   // the merge over the ranks (+ the worst status)
        int np = 0;
        int local = FIR_OK;
        char mine_err[512] = "";
        for (size_t pi = 0; pi < h->parts.size() && !local; ++pi) {
            auto& p = h->parts[pi];
            if (p.slot != slot) continue;
            if (cls_inject(h, (int)pi, 1)) { local = sh_fail(FIR_ERR_NOMEM, "injected scan failure on training-set shard %d (fir_shard_opts.fail_step = 1)", (int)pi); break; }
            if (!p.c) continue;
            double* dl = nullptr;
            void* st = nullptr;
            local = fir_cls_knn_nearest_dev_(p.c, queries, qb, k, &dl, &st, nullptr);
            if (local) break;
            if (hipStreamSynchronize((hipStream_t)st) != hipSuccess ||
                hipMemcpyAsync(parts + per * (size_t)np, dl, per * sizeof(double), hipMemcpyDeviceToDevice, dc.stream) != hipSuccess)
                local = sh_fail(FIR_ERR_HIP, "copying a shard's k-nearest table failed");
            ++np;
        }
        if (local) { strncpy(mine_err, fir_last_error(), sizeof(mine_err) - 1); (void)hipGetLastError(); np = 0; }
        hipLaunchKernelGGL(k_shard_merge_knn, dim3((unsigned)((qc + 255) / 256)), dim3(256), 0, dc.stream, parts, np, qc, k, mine, per, (double*)nullptr);   // np == 0: all DBL_MAX
        hipLaunchKernelGGL(k_shard_set_f64, dim3(1), dim3(1), 0, dc.stream, mine + per, local ? (double)(-local) : 0.0);
        const ncclResult_t nr = ncclAllGather(mine, gath, per + 1, ncclDouble, dc.comm, dc.stream);
        if (nr != ncclSuccess) return comm_dead(dc, h->health, FIR_ERR_COMM, ncclGetErrorString(nr));
        hipLaunchKernelGGL(k_shard_merge_knn, dim3((unsigned)((qc + 255) / 256)), dim3(256), 0, dc.stream, gath, h->nranks, qc, k, all, per + 1, all + per);
        hipLaunchKernelGGL(k_shard_knn_vote, dim3((qb + 63) / 64), dim3(64), 0, dc.stream, all, qb, h->num_classes, k, h->class_count[(size_t)slot],
                           h->best[(size_t)slot]);
        double* hst = slot == 0 ? (double*)((char*)h->pin + (((size_t)qb * sizeof(int32_t) + 7) & ~(size_t)7)) : (double*)(dc.h_status + 4);
        (void)hipMemcpyAsync(hst, all + per, sizeof(double), hipMemcpyDeviceToHost, dc.stream);
        if (slot == 0) (void)hipMemcpyAsync(h->pin, h->best[0], (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, dc.stream);
        if ((r = wait_stream(dc, dc.stream, h->health))) return r;
        if (local) { h->health.dead = true; h->health.dead_code = local; fir_set_last_error_(mine_err); return local; }
        if (*hst != 0.0) { h->health.dead = true; h->health.dead_code = -(int)*hst; return peer_failed(-(int)*hst); }
        return FIR_OK;
    });
    if (grows && !h->health.dead) { h->knn_cap = std::max(h->knn_cap, need); h->best_cap = std::max(h->best_cap, new_best); }
    if (rc) { h->health.dead = true; if (!h->health.dead_code.load()) h->health.dead_code = rc; return rc; }
    std::memcpy(best_class, h->pin, (size_t)qb * sizeof(int32_t));
    return FIR_OK;
}

int fir_cls_sharded_pnn_predict(fir_cls_sharded* h, const double* queries, int32_t qb, double var, double* scores, int32_t* best_class) {
    if (!h || (qb > 0 && !queries)) return sh_fail(FIR_ERR_ARG, "NULL argument");
    if (qb < 0) return sh_fail(FIR_ERR_ARG, "qb < 0");
    if (h->health.dead) return dead_handle(h->health);
    if (qb == 0) return FIR_OK;
    // internal batches: what every shard's distance table allows (fir_cls_pnn_scores_dev_)
    int32_t maxb = 1 << 30;
    for (auto& p : h->parts)
        if (p.c) { int32_t mb = 0; double* ds; void* st; fir_cls_pnn_scores_dev_(p.c, nullptr, 0, var, &ds, &st, &mb); maxb = std::min(maxb, mb); }
    maxb = std::min(maxb, 4096);
    if (qb > maxb) {
        for (int32_t q0 = 0; q0 < qb; q0 += maxb) {
            const int rc0 = fir_cls_sharded_pnn_predict(h, queries + (size_t)q0 * h->d, std::min(maxb, qb - q0), var,
                                                        scores ? scores + (size_t)q0 * h->num_classes : nullptr, best_class ? best_class + q0 : nullptr);
            if (rc0) return rc0;
        }
        return FIR_OK;
    }
    const size_t nsc = (size_t)qb * h->num_classes;
    const size_t pin_need = (nsc + 1) * sizeof(double) + (size_t)qb * sizeof(int32_t);
    const bool grows = nsc + 1 > h->acc_cap || (size_t)qb > h->best_cap || pin_need > h->pin_cap;      // the same answer on every rank
    const size_t new_best = std::max<size_t>(4096, (size_t)qb);
    int rc = run_all(h->devs, [&](int slot) -> int {
        DevCtx& dc = h->devs[(size_t)slot];
        SH_HIP(hipSetDevice(dc.device));
        int r = grow_agreed(dc, dc.stream, h->health, grows, cls_inject_dev(h, slot, 2), [&]() -> int {
            if (nsc + 1 > h->acc_cap) {
                if (h->acc[(size_t)slot]) SH_HIP(hipFree(h->acc[(size_t)slot]));
                h->acc[(size_t)slot] = nullptr;
                SH_HIP(hipMalloc((void**)&h->acc[(size_t)slot], (nsc + 1) * sizeof(double)));
            }
            if ((size_t)qb > h->best_cap) {
                if (h->best[(size_t)slot]) SH_HIP(hipFree(h->best[(size_t)slot]));
                h->best[(size_t)slot] = nullptr;
                SH_HIP(hipMalloc((void**)&h->best[(size_t)slot], new_best * sizeof(int32_t)));
            }
            if (slot == 0 && pin_need > h->pin_cap) {
                if (h->pin) SH_HIP(hipHostFree(h->pin));
                h->pin = nullptr; h->pin_cap = 0;
                SH_HIP(hipHostMalloc(&h->pin, pin_need, hipHostMallocPortable));
                h->pin_cap = pin_need;
            }
            return FIR_OK;
        });
        if (r) return r;
        double* acc = h->acc[(size_t)slot];
        int first = 1;
        int local = FIR_OK;
        char mine_err[512] = "";
        for (size_t pi = 0; pi < h->parts.size() && !local; ++pi) {
            auto& p = h->parts[pi];
            if (p.slot != slot) continue;
            if (cls_inject(h, (int)pi, 1)) { local = sh_fail(FIR_ERR_NOMEM, "injected scan failure on training-set shard %d (fir_shard_opts.fail_step = 1)", (int)pi); break; }
            if (!p.c) continue;
            double* ds = nullptr;
            void* st = nullptr;
            local = fir_cls_pnn_scores_dev_(p.c, queries, qb, var, &ds, &st, nullptr);
            if (local) break;
            if (hipStreamSynchronize((hipStream_t)st) != hipSuccess) { local = sh_fail(FIR_ERR_HIP, "a shard's class scores did not complete"); break; }   // the shard's own stream
            hipLaunchKernelGGL(k_shard_add_f64, dim3((unsigned)((nsc + 255) / 256)), dim3(256), 0, dc.stream, acc, ds, (int64_t)nsc, first);
            first = 0;
        }
        if (local) { strncpy(mine_err, fir_last_error(), sizeof(mine_err) - 1); (void)hipGetLastError(); }
        if (first || local) (void)hipMemsetAsync(acc, 0, nsc * sizeof(double), dc.stream);     // a device without rows (or whose step failed) adds nothing
        hipLaunchKernelGGL(k_shard_set_f64, dim3(1), dim3(1), 0, dc.stream, acc + nsc, local ? 1.0 : 0.0);   // the status element: the sum counts the ranks that failed
        const ncclResult_t nr = ncclAllReduce(acc, acc, nsc + 1, ncclDouble, ncclSum, dc.comm, dc.stream);
        if (nr != ncclSuccess) return comm_dead(dc, h->health, FIR_ERR_COMM, ncclGetErrorString(nr));
        hipLaunchKernelGGL(k_shard_argmax_first, dim3(qb), dim3(64), 0, dc.stream, acc, h->num_classes, h->best[(size_t)slot]);
        double* hst = slot == 0 ? (double*)h->pin + nsc : (double*)(dc.h_status + 4);
        if (slot == 0) {
            (void)hipMemcpyAsync(h->pin, acc, (nsc + 1) * sizeof(double), hipMemcpyDeviceToHost, dc.stream);
            (void)hipMemcpyAsync((char*)h->pin + (nsc + 1) * sizeof(double), h->best[0], (size_t)qb * sizeof(int32_t), hipMemcpyDeviceToHost, dc.stream);
        } else {
            (void)hipMemcpyAsync(hst, acc + nsc, sizeof(double), hipMemcpyDeviceToHost, dc.stream);
        }
        if ((r = wait_stream(dc, dc.stream, h->health))) return r;
        if (local) { h->health.dead = true; h->health.dead_code = local; fir_set_last_error_(mine_err); return local; }
        if (*hst != 0.0) { h->health.dead = true; h->health.dead_code = FIR_ERR_COMM; return peer_failed(FIR_ERR_COMM); }
        return FIR_OK;
    });
    if (grows && !h->health.dead) { h->acc_cap = std::max(h->acc_cap, nsc + 1); h->best_cap = std::max(h->best_cap, new_best); }
    if (rc) { h->health.dead = true; if (!h->health.dead_code.load()) h->health.dead_code = rc; return rc; }
    if (scores) std::memcpy(scores, h->pin, nsc * sizeof(double));
    if (best_class) std::memcpy(best_class, (char*)h->pin + (nsc + 1) * sizeof(double), (size_t)qb * sizeof(int32_t));
    return FIR_OK;
}

int fir_comm_unique_id(void* id_out) {
    if (!id_out) return sh_fail(FIR_ERR_ARG, "id_out is NULL");
    static_assert(sizeof(ncclUniqueId) <= FIR_COMM_ID_BYTES, "FIR_COMM_ID_BYTES too small");
    ncclUniqueId id;
    SH_NCCL(ncclGetUniqueId(&id));
    std::memset(id_out, 0, FIR_COMM_ID_BYTES);
    std::memcpy(id_out, &id, sizeof id);
    return FIR_OK;
}

int fir_sharded_destroy(fir_sharded* h) {
    if (!h) return FIR_OK;
    for (Shard& s : h->shards)
        if (s.g) fir_gallery_destroy(s.g);
    teardown_devices(h->devs, h->health.dead);
    if (h->pin) (void)hipHostFree(h->pin);
    delete h;
    return FIR_OK;
}

int fir_gallery_create_sharded_ex(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric, const int32_t* devices,
                                  int32_t ndev, const fir_shard_opts* opts, fir_sharded** out) {
    if (!out) return sh_fail(FIR_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || ndev < 1 || ndev > 64) return sh_fail(FIR_ERR_ARG, "device list of %d entries", ndev);
    if (n < 0 || d <= 0 || (n > 0 && !rows)) return sh_fail(FIR_ERR_ARG, "bad gallery shape n=%lld d=%d", (long long)n, d);
    fir_shard_opts o;
    std::memset(&o, 0, sizeof o);
    if (opts) {
        if (opts->struct_bytes < 8 || opts->struct_bytes > (int32_t)sizeof o) return sh_fail(FIR_ERR_ARG, "fir_shard_opts.struct_bytes = %d", opts->struct_bytes);
        std::memcpy(&o, opts, (size_t)opts->struct_bytes);
    }
    const int spd = o.shards_per_device > 0 ? o.shards_per_device : 1;
    const int nprocs = o.comm_id ? o.nprocs : 1, proc = o.comm_id ? o.proc_rank : 0;
    if (spd > 64) return sh_fail(FIR_ERR_ARG, "shards_per_device = %d", spd);
    if (nprocs < 1 || proc < 0 || proc >= nprocs) return sh_fail(FIR_ERR_ARG, "process %d of %d", proc, nprocs);
    if (o.rows_on_device && ndev != 1) return sh_fail(FIR_ERR_ARG, "device-resident rows need a one-entry device list (they live on one device)");
#ifndef FIR_AUDIT
    if (o.fail_shard != 0 || o.fail_step != 0)       // (as read_opts: the hooks exist in the audit build only)
        return sh_fail(FIR_ERR_ARG, "fir_shard_opts.fail_shard / fail_step are test hooks of the audit build (libfir_amd_audit.so); this library has none");
#endif
    if (o.first_global_row < 0 || o.first_global_row + n >= ((int64_t)1 << 31)) return sh_fail(FIR_ERR_ARG, "rows [%lld, +%lld) do not fit 32-bit indices", (long long)o.first_global_row, (long long)n);
    for (int i = 0; i < ndev; ++i)
        for (int j = 0; j < i; ++j)
            if (devices[i] == devices[j]) return sh_fail(FIR_ERR_ARG, "device %d listed twice (use shards_per_device for logical shards)", devices[i]);
    const int cnt = fir_device_count();
    for (int i = 0; i < ndev; ++i)
        if (devices[i] < 0 || devices[i] >= cnt) return sh_fail(FIR_ERR_NODEVICE, "device %d out of range (%d visible)", devices[i], cnt);

    fir_sharded* h = new (std::nothrow) fir_sharded();
    if (!h) return sh_fail(FIR_ERR_NOMEM, "host allocation failed");
    h->d = d; h->metric = metric; h->n_local = n; h->first_row = o.first_global_row;
    h->ndev = ndev; h->spd = spd; h->nranks = nprocs * ndev; h->rank0 = proc * ndev;
    h->has_labels = class_no != nullptr;
    h->health.apply(o);
    if (h->health.fail_shard < 0 || h->health.fail_shard > ndev * spd) { delete h; return sh_fail(FIR_ERR_ARG, "fail_shard = %d of %d shards", o.fail_shard, ndev * spd); }
    h->devs.resize((size_t)ndev);
    int rc = FIR_OK;
    auto bail = [&](int code) { fir_sharded_destroy(h); return code; };

    // shards: whole 64-row tiles, contiguous, device-major (device i holds shards i*spd .. i*spd+spd-1)
    const int nsh = ndev * spd;
    const int64_t tiles = (n + fir::kTileRows - 1) / fir::kTileRows;
    const int64_t per = ((tiles + nsh - 1) / nsh) * fir::kTileRows;
    h->shards.resize((size_t)nsh);
    for (int s = 0; s < nsh && rc == FIR_OK; ++s) {
        Shard& sh = h->shards[(size_t)s];
        sh.slot = s / spd;
        const int64_t lo = std::min<int64_t>((int64_t)s * per, n), hi = std::min<int64_t>((int64_t)(s + 1) * per, n);
        sh.lo = o.first_global_row + lo;
        sh.hi = o.first_global_row + hi;
        h->devs[(size_t)sh.slot].shards.push_back(s);
        if (hi <= lo) continue;
        const int dev = devices[sh.slot];
        if (o.rows_on_device) rc = fir_gallery_create_dev(rows + lo * d, hi - lo, d, class_no ? class_no + lo : nullptr, metric, dev, nullptr, &sh.g);
        else rc = fir_gallery_create(rows + lo * d, hi - lo, d, class_no ? class_no + lo : nullptr, metric, dev, &sh.g);
        if (rc == FIR_OK) rc = fir_gallery_set_row_offset(sh.g, sh.lo);
        if (rc == FIR_OK) {
            fir_gallery_view v;
            fir_gallery_view_(sh.g, &v);
            sh.cls = v.cls;
        }
    }
    if (rc) return bail(rc);

    // per-device stream, communicator (rank = proc * ndev + slot), worker threads
    if ((rc = setup_devices(h->devs, devices, ndev, h->nranks, h->rank0, o.comm_id))) return bail(rc);
    *out = h;
    return FIR_OK;
}

int fir_gallery_create_sharded(const float* rows, int64_t n, int32_t d, const int32_t* class_no, int32_t metric, const int32_t* devices,
                               int32_t ndev, fir_sharded** out) {
    return fir_gallery_create_sharded_ex(rows, n, d, class_no, metric, devices, ndev, nullptr, out);
}

int fir_sharded_info(const fir_sharded* h, int64_t* n_local, int32_t* d, int32_t* ndev, int32_t* nshards, int32_t* nranks, int32_t* first_rank) {
    if (!h) return sh_fail(FIR_ERR_ARG, "handle is NULL");
    if (n_local) *n_local = h->n_local;
    if (d) *d = h->d;
    if (ndev) *ndev = h->ndev;
    if (nshards) *nshards = (int32_t)h->shards.size();
    if (nranks) *nranks = h->nranks;
    if (first_rank) *first_rank = h->rank0;
    return FIR_OK;
}

int fir_sharded_shard(fir_sharded* h, int32_t i, fir_gallery** g, int64_t* first_global_row, int64_t* rows) {
    if (!h || i < 0 || i >= (int32_t)h->shards.size()) return sh_fail(FIR_ERR_ARG, "shard %d of %d", i, h ? (int)h->shards.size() : 0);
    if (g) *g = h->shards[(size_t)i].g;
    if (first_global_row) *first_global_row = h->shards[(size_t)i].lo;
    if (rows) *rows = h->shards[(size_t)i].hi - h->shards[(size_t)i].lo;
    return FIR_OK;
}

int fir_sharded_set_metric(fir_sharded* h, int32_t metric) {
    if (!h) return sh_fail(FIR_ERR_ARG, "handle is NULL");
    for (Shard& s : h->shards)
        if (s.g) { const int rc = fir_gallery_set_metric(s.g, metric); if (rc) return rc; }
    h->metric = metric;
    return FIR_OK;
}

int fir_sharded_search_top1(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t* idx, float* dist) {
    return search_host(h, queries, qb, start_pos, end_pos, 1, idx, dist, nullptr);
}

int fir_sharded_search_topk(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t k, int32_t* idx,
                            float* dist) {
    return search_host(h, queries, qb, start_pos, end_pos, k, idx, dist, nullptr);
}

int fir_sharded_classify_top1(fir_sharded* h, const float* queries, int32_t qb, int32_t start_pos, int32_t end_pos, int32_t* class_out,
                              int32_t* idx, float* dist) {
    if (!class_out) return sh_fail(FIR_ERR_ARG, "class_out is NULL");
    return search_host(h, queries, qb, start_pos, end_pos, 1, idx, dist, class_out);
}

int fir_sharded_search_top1_keys_dev(fir_sharded* h, const float* d_queries, int32_t qb, int32_t start_pos, int32_t end_pos, uint64_t* d_keys,
                                     void* stream) {
    if (!h || !d_keys || (qb > 0 && !d_queries)) return sh_fail(FIR_ERR_ARG, "NULL argument");
    if (h->ndev != 1) return sh_fail(FIR_ERR_STATE, "device-pointer calls need a one-device handle (one process per GPU); this one lists %d", h->ndev);
    if (qb < 0) return sh_fail(FIR_ERR_ARG, "qb < 0");
    if (h->health.dead) return dead_handle(h->health);
    if (qb == 0) return FIR_OK;
    int rc = check_range(h, start_pos, end_pos);
    if (rc) return rc;
    DevCtx& dc = h->devs[0];
    SH_HIP(hipSetDevice(dc.device));
    hipStream_t st = stream ? (hipStream_t)stream : dc.stream;
    // What the previous asynchronous call found, if it is through (no waiting here): a peer's failure closes the handle BEFORE this rank
    // enqueues another collective that the failed peer -- dead since that call -- will never enter. A previous call that is still in
    // flight is the caller's to order (fir_amd.h: fir_sharded_sync between asynchronous calls whose failure must not be outrun).
    if (dc.async_pending && hipEventQuery(dc.async_done) == hipSuccess) {
        dc.async_pending = false;
        if (dc.h_status[3] < 0) { h->health.dead = true; h->health.dead_code = dc.h_status[3]; return peer_failed(dc.h_status[3]); }
    }
    (void)hipGetLastError();
    const CallBuffers nb = call_buffers(h, dc, qb, 1, false, false);
    if ((rc = grow_agreed(dc, st, h->health, call_grows(dc, nb), inject_here(h, dc, 2), [&]() -> int { return call_alloc(dc, nb); }))) return rc;
    // every rank enters the exchange; one whose scans failed sends FIR_KEY_NONE keys and a poisoned status element, returns its
    // error here, and the others find it in fir_sharded_sync (the call is asynchronous: nobody waits for the exchange here)
    int local = device_keys(h, 0, d_queries, qb, start_pos, end_pos, 1, st);
    char mine[512] = "";
    if (local) { strncpy(mine, fir_last_error(), sizeof(mine) - 1); (void)hipGetLastError(); }
    if ((rc = exchange_keys(h, 0, qb, 1, st, local))) return rc;
    hipLaunchKernelGGL(k_shard_note_status, dim3(1), dim3(1), 0, st, dc.keys + qb, dc.status + 1, dc.d_hsticky);
    (void)hipMemcpyAsync(d_keys, dc.keys, (size_t)qb * sizeof(uint64_t), hipMemcpyDeviceToDevice, st);
    (void)hipEventRecord(dc.async_done, st);           // what fir_sharded_sync (and the next call's look at the status) waits for: `st` may be the caller's stream
    dc.async_pending = true;
    if (local) { h->health.dead = true; h->health.dead_code = local; fir_set_last_error_(mine); return local; }
    SH_HIP(hipGetLastError());
    return FIR_OK;
}

int fir_sharded_sync(fir_sharded* h) {
    if (!h) return sh_fail(FIR_ERR_ARG, "handle is NULL");
    if (h->health.dead) return dead_handle(h->health);
    for (DevCtx& dc : h->devs) {
        SH_HIP(hipSetDevice(dc.device));
        int rc;
        // the last asynchronous call ran on whatever stream its caller named: its completion is the event recorded behind it, waited for
        // with the same bound as every other wait behind a collective (the handle's own stream says nothing about it)
        if (dc.async_pending && (rc = wait_event(dc, dc.async_done, h->health))) return rc;
        dc.async_pending = false;
        if ((rc = wait_stream(dc, dc.stream, h->health))) return rc;
        // the asynchronous calls' record: the worst status any exchange since the last sync came back with (device word and its host mirror)
        SH_HIP(hipMemcpyAsync(dc.h_status + 2, dc.status + 1, sizeof(int32_t), hipMemcpyDeviceToHost, dc.stream));
        if ((rc = wait_stream(dc, dc.stream, h->health))) return rc;
        const int32_t worst = std::min(dc.h_status[2], dc.h_status[3]);
        if (worst < 0) { h->health.dead = true; h->health.dead_code = worst; return peer_failed(worst); }
    }
    return FIR_OK;
}

int fir_sharded_profile_enable(fir_sharded* h, int32_t on) {
    if (!h) return sh_fail(FIR_ERR_ARG, "handle is NULL");
    h->profiling = on != 0;
    h->devs[0].ev_used = 0;
    return FIR_OK;
}

int fir_sharded_profile_read(fir_sharded* h, float* exchange_ms, int32_t cap, int32_t* count) {
    if (!h) return sh_fail(FIR_ERR_ARG, "handle is NULL");
    DevCtx& dc = h->devs[0];
    SH_HIP(hipSetDevice(dc.device));
    const int32_t have = (int32_t)(dc.ev_used / 2);
    for (int32_t i = 0; i < have; ++i) {
        SH_HIP(hipEventSynchronize(dc.evs[2 * (size_t)i + 1]));
        float ms = 0.f;
        SH_HIP(hipEventElapsedTime(&ms, dc.evs[2 * (size_t)i], dc.evs[2 * (size_t)i + 1]));
        if (exchange_ms && i < cap) exchange_ms[i] = ms;
    }
    if (count) *count = have;
    dc.ev_used = 0;
    return FIR_OK;
}

}  // extern "C"
