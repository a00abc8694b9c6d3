"""Row-sharded gallery behind the C ABI (fir_gallery_create_sharded*, include/fir_amd.h): the split, the per-shard
scans, the on-device minimum over a device's shards and the RCCL exchange (ncclAllReduce(min, u64) / ncclAllGather /
ncclAllReduce(min, i32)) all run inside libfir_amd.so. On a one-GPU box the device list is [0] and the gallery is cut
into logical shards; the communicator then has one rank, but every call on the path is the one an 8-GPU node makes.

Bar: identical to the unsharded handle and to the oracle -- index, distance bits, first-minimum tie-break across
shard borders (qt_cpp/db_features.cpp:329-332), -1 / 100000 when nothing qualifies (:322-323)."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

L2, CHI2, KL = 0, 1, 2


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("n,d,spd", [(4099, 512, 8), (1000, 256, 3), (64, 128, 8), (130, 64, 2), (70000, 64, 8)])
def test_top1_over_logical_shards_equals_unsharded_and_oracle(fir, oracle, n, d, spd):
    rows = synth.make_gallery(29, n, d, L2)
    q, _ = synth.make_queries(29, rows, 21, L2)
    rows[n - 2] = rows[3]
    q[1] = rows[3]                       # an exact tie between a row of the first shard and one of the last
    q[2] = np.float32(1e4)               # nothing below 100000 in any shard
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        idx0, dist0 = g.search_top1(q)
    with fir.ShardedGallery(rows, None, fir.METRIC_L2, devices=[0], shards_per_device=spd) as s:
        info = s.info()
        assert info["nshards"] == spd and info["nranks"] == 1 and info["ndev"] == 1
        idx, dist = s.search_top1(q)
        # the shards tile the rows: contiguous, whole 64-row tiles, nothing lost
        covered = 0
        for i in range(spd):
            gal, lo, cnt = s.shard(i)
            assert lo == covered or cnt == 0
            assert (cnt % 64 == 0) or lo + cnt == n or cnt == 0
            covered += cnt
        assert covered == n
    assert np.array_equal(idx, idx0) and np.array_equal(bits(dist), bits(dist0))
    eidx, edist = oracle.top1_batch(rows, q, 0, d, L2)
    assert np.array_equal(idx, eidx) and np.array_equal(bits(dist), bits(edist))
    assert idx[1] == 3 and idx[2] == -1 and dist[2] == np.float32(100000.0)


@pytest.mark.parametrize("metric", [L2, CHI2])
def test_topk_allgather_merge_and_feature_subrange(fir, oracle, metric):
    n, d, k = 3000, 256, 5
    rows = synth.make_gallery(31, n, d, metric)
    q, _ = synth.make_queries(31, rows, 9, metric)
    rows[2900] = rows[10]
    rows[1500] = rows[10]
    q[0] = rows[10]                      # three equal distances in three different shards: ordered by global row index
    with fir.ShardedGallery(rows, None, metric, devices=[0], shards_per_device=4) as s:
        idx, dist = s.search_topk(q, k, 0, 64)
        idx1, dist1 = s.search_top1(q, 32, 200)
    for i in range(q.shape[0]):
        ei, ed = oracle.topk(rows, q[i], 0, 64, k, metric)
        assert np.array_equal(idx[i], ei), (i, idx[i], ei)
        assert np.array_equal(bits(dist[i]), bits(ed))
    assert list(idx[0][:3]) == [10, 1500, 2900]
    eidx, edist = oracle.top1_batch(rows, q, 32, 200, metric)
    assert np.array_equal(idx1, eidx) and np.array_equal(bits(dist1), bits(edist))


def test_classify_top1_is_bruteforce_classifier(fir, oracle):
    """BruteForceClassifier::recognize (ImageTesting.cpp:58-71) over shards: classNo of the nearest row, -1 when none."""
    n, d = 5000, 128
    rows = synth.make_gallery(37, n, d, L2)
    labels = synth.make_labels(n, 101)
    q, _ = synth.make_queries(37, rows, 40, L2)
    q[5] = np.float32(1e4)
    with fir.ShardedGallery(rows, labels, fir.METRIC_L2, devices=[0], shards_per_device=5) as s:
        cls, idx, dist = s.classify_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, d, L2)
    assert np.array_equal(idx, eidx) and np.array_equal(bits(dist), bits(edist))
    exp = np.where(eidx >= 0, labels[np.maximum(eidx, 0)], -1)
    assert np.array_equal(cls, exp)
    assert cls[5] == -1
    with fir.ShardedGallery(rows, None, fir.METRIC_L2, devices=[0]) as s:
        with pytest.raises(fir.FirError):
            s.classify_top1(q)


def test_device_pointer_form_and_large_batch_through_matrix_cores(fir, oracle):
    """One-process-per-GPU form (device pointers, caller's stream) with a batch large enough for the library's automatic
    matrix-core dispatch in every shard; the exchange is profiled."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, d, qb = 40000, 512, 256
    rows = synth.make_gallery(41, n, d, L2)
    q, _ = synth.make_queries(41, rows, qb, L2)
    rt = torch.from_numpy(rows).to(dev)
    qt = torch.from_numpy(q).to(dev)
    keys = torch.empty(qb, device=dev, dtype=torch.int64)
    st = torch.cuda.Stream(device=dev)
    with fir.ShardedGallery(dev_ptr=rt.data_ptr(), n=n, d=d, metric=fir.METRIC_L2, devices=[0], shards_per_device=8, first_global_row=1000) as s:
        s.profile_enable(True)
        with torch.cuda.stream(st):
            for _ in range(3):
                s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
        ex = s.profile_read()
        assert ex.size == 3 and np.all(ex >= 0)
    idx, dist = fir.keys_unpack(keys.cpu().numpy().view(np.uint64))
    eidx, edist = oracle.top1_batch(rows, q[:48], 0, d, L2)
    assert np.array_equal(idx[:48], eidx + 1000) and np.array_equal(bits(dist[:48]), bits(edist))
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        idx0, dist0 = g.search_top1(q)
    assert np.array_equal(idx, idx0 + 1000) and np.array_equal(bits(dist), bits(dist0))


def test_argument_errors(fir):
    rows = synth.make_gallery(1, 100, 16, L2)
    with pytest.raises(fir.FirError):
        fir.ShardedGallery(rows, None, 0, devices=[0, 0])              # a device listed twice
    with pytest.raises(fir.FirError):
        fir.ShardedGallery(rows, None, 0, devices=[fir.device_count()])
    with pytest.raises(fir.FirError):
        fir.ShardedGallery(rows, None, 0, devices=[])
    with fir.ShardedGallery(rows, None, 0, devices=[0], shards_per_device=4) as s:   # 100 rows = 2 tiles: two shards stay empty
        idx, _ = s.search_top1(rows[:3])
        assert list(idx) == [0, 1, 2]
        with pytest.raises(fir.FirError):
            s.search_top1(rows[:3], 4, 2)
        with pytest.raises(fir.FirError):
            s.search_topk(rows[:3], 9)


@pytest.mark.parametrize("step", [1, 2])
def test_a_failing_shard_fails_the_call_on_every_entry_point_and_never_blocks(fir, fir_audit, oracle, step):
    """Failure semantics (include/fir_amd.h, 'Failure semantics of every sharded handle'; the reference's "-1, never block",
    ann.cpp:113-126): shard 3 of 8 is made to fail -- its scan after the buffers were agreed on (step 1: the rank still enters
    the exchange, with FIR_KEY_NONE keys and a poisoned status element) or its device's buffer growth (step 2: the one-int
    status all-reduce of the agreed growth step). The call returns the shard's error code instead of hanging, the handle is
    closed (FIR_ERR_STATE at once), destroying it works, and a fresh handle answers like the unsharded one."""
    n, d = 4099, 128
    rows = synth.make_gallery(57, n, d, L2)
    labels = synth.make_labels(n, 17)
    q, _ = synth.make_queries(57, rows, 11, L2)
    calls = [lambda s: s.search_top1(q), lambda s: s.search_topk(q, 5), lambda s: s.classify_top1(q)]
    for call in calls:
        # (the injection hooks live in the audit build only: libfir_amd_audit.so)
        with fir_audit.ShardedGallery(rows, labels, fir.METRIC_L2, devices=[0], shards_per_device=8, fail_shard=4, fail_step=step, timeout_ms=20000) as s:
            with pytest.raises(fir_audit.FirError) as e:
                call(s)
            assert e.value.code == -3, (e.value.code, str(e.value))            # FIR_ERR_NOMEM, the injected error
            for again in calls:                                               # closed: no collective is attempted any more
                with pytest.raises(fir_audit.FirError) as e2:
                    again(s)
                assert e2.value.code == -5                                    # FIR_ERR_STATE
    with pytest.raises(fir.FirError) as e3:                                   # the shipped library has no such hook and says so
        fir.ShardedGallery(rows, labels, fir.METRIC_L2, devices=[0], shards_per_device=8, fail_shard=4, fail_step=step)
    assert e3.value.code == -1
    with fir.ShardedGallery(rows, labels, fir.METRIC_L2, devices=[0], shards_per_device=8) as s:     # a fresh handle works
        idx, dist = s.search_top1(q)
        cls, _, _ = s.classify_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, d, L2)
    assert np.array_equal(idx, eidx) and np.array_equal(bits(dist), bits(edist))
    assert np.array_equal(cls, np.where(eidx >= 0, labels[np.maximum(eidx, 0)], -1))


def test_a_failing_shard_in_the_asynchronous_device_pointer_call(fir, fir_audit):
    """fir_sharded_search_top1_keys_dev returns before the exchange has run: the failing rank gets its error from the call, the
    exchange still happens (nobody is left waiting in it), and fir_sharded_sync reports the closed handle."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, d, qb = 5000, 64, 16
    rows = synth.make_gallery(58, n, d, L2)
    q, _ = synth.make_queries(58, rows, qb, L2)
    rt, qt = torch.from_numpy(rows).to(dev), torch.from_numpy(q).to(dev)
    keys = torch.empty(qb, device=dev, dtype=torch.int64)
    with fir_audit.ShardedGallery(dev_ptr=rt.data_ptr(), n=n, d=d, metric=fir.METRIC_L2, devices=[0], shards_per_device=4, fail_shard=2, fail_step=1) as s:
        with pytest.raises(fir_audit.FirError) as e:
            s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr())
        assert e.value.code == -3
        with pytest.raises(fir_audit.FirError) as e2:
            s.sync()
        assert e2.value.code == -5
    with fir.ShardedGallery(dev_ptr=rt.data_ptr(), n=n, d=d, metric=fir.METRIC_L2, devices=[0], shards_per_device=4) as s:
        s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr())
        s.sync()
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        idx0, _ = g.search_top1(q)
    idx, _ = fir.keys_unpack(keys.cpu().numpy().view(np.uint64))
    assert np.array_equal(idx, idx0)


def test_a_peers_failure_reaches_the_asynchronous_caller_whatever_stream_it_used(fir_audit):
    """the asynchronous device-pointer call may run on the CALLER's stream; fir_sharded_sync must wait for THAT
    work (an event recorded behind the call, with the handle's time-out) before it reads the sticky status, and the next call looks
    at the status on entry. fail_step = 3 (audit build) poisons the status element that comes back from the handle's second exchange
    on -- this rank's own scan is fine, it is a peer that 'failed': the call itself succeeds, the sync reports FIR_ERR_NOMEM and
    closes the handle; with a sleeping kernel in front of the call on the caller's stream the sync still waits for it."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, d, qb = 6000, 64, 16
    rows = synth.make_gallery(59, n, d, L2)
    q, _ = synth.make_queries(59, rows, qb, L2)
    rt, qt = torch.from_numpy(rows).to(dev), torch.from_numpy(q).to(dev)
    keys = torch.empty(qb, device=dev, dtype=torch.int64)
    torch.cuda.synchronize()
    mine = torch.cuda.Stream()                              # a non-blocking stream of the caller's own
    opts = dict(dev_ptr=rt.data_ptr(), n=n, d=d, metric=fir_audit.METRIC_L2, devices=[0], shards_per_device=4, fail_shard=2, fail_step=3, timeout_ms=20000)
    with fir_audit.ShardedGallery(**opts) as s:
        s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)      # clean (and it grows the buffers: blocking)
        s.sync()
        with torch.cuda.stream(mine):
            torch.cuda._sleep(400_000_000)                  # ~0.2 s of work in front of the call on the caller's stream
            s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)  # succeeds: nothing of THIS rank failed
        assert not mine.query()                             # (the call did not wait)
        with pytest.raises(fir_audit.FirError) as e:
            s.sync()
        assert e.value.code == -3, (e.value.code, str(e.value))   # the peer's error, found on the caller's stream
        assert mine.query()                                 # the sync waited for the caller's stream's work
        with pytest.raises(fir_audit.FirError) as e2:       # closed
            s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)
        assert e2.value.code == -5
    # and without a sync in between: the next call finds the status on entry once the poisoned call's work is through
    with fir_audit.ShardedGallery(**opts) as s:
        s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)
        s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)      # the poisoned exchange
        mine.synchronize()
        with pytest.raises(fir_audit.FirError) as e3:
            s.search_top1_keys_dev(qt.data_ptr(), qb, keys.data_ptr(), stream=mine.cuda_stream)
        assert e3.value.code == -3
