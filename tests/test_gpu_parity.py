"""GPU parity: the HIP path (through the C ABI) against the oracle on the same seeded inputs.

Bar: bit-exact index AND distance for L2 and chi-square (integer-exact IEEE arithmetic in the
reference's evaluation order); KL within 1e-5 relative (the reference calls libm logf, the
device its own logf) with identical top-1 wherever the runner-up is further than that.
"""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

L2, CHI2, KL = 0, 1, 2


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bits_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    bad = np.nonzero(bits(a) != bits(b))[0]
    assert bad.size == 0, f"{what}: {bad.size} mismatches, first at {bad[:5]}: {a.ravel()[bad[:5]]} vs {b.ravel()[bad[:5]]}"


@pytest.mark.parametrize("seed,n,d", [(13, 257, 64), (17, 1000, 256), (101, 4099, 512), (13, 1000, 1280), (17, 700, 1536),
                                      (101, 64, 512), (13, 65, 100), (17, 1, 7), (5, 8191, 33)])
def test_l2_top1_bit_exact(fir, oracle, seed, n, d):
    rows = synth.make_gallery(seed, n, d, L2)
    q, _ = synth.make_queries(seed, rows, 11, L2)
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, d, L2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist, "best distance")


@pytest.mark.parametrize("qb", [1, 2, 3, 5, 8, 9, 16, 31, 64])
def test_l2_top1_batch_sizes(fir, oracle, qb):
    rows = synth.make_gallery(3, 2000, 128, L2)
    q, _ = synth.make_queries(3, rows, qb, L2)
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, 128, L2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)


@pytest.mark.parametrize("qpp", [1, 2, 4, 8])
def test_l2_top1_queries_per_pass(fir, oracle, qpp):
    rows = synth.make_gallery(4, 3000, 512, L2)
    q, _ = synth.make_queries(4, rows, 10, L2)
    with fir.Gallery(rows, None, fir.METRIC_L2, 0) as g:
        g.set_tuning(queries_per_pass=qpp)
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, 512, L2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)


@pytest.mark.parametrize("start,end", [(0, 64), (0, 256), (64, 256), (32, 64), (0, 1), (5, 7), (3, 250), (4, 7), (5, 8), (1, 2),
                                       (0, 300), (17, 299), (255, 256)])
@pytest.mark.parametrize("metric", [L2, CHI2])
def test_feature_subranges(fir, oracle, start, end, metric):
    """[start,end) semantics of ImageTesting.cpp:117,174,243 incl. ranges not aligned to 4."""
    rows = synth.make_gallery(7, 777, 300, metric)
    q, _ = synth.make_queries(7, rows, 5, metric)
    with fir.Gallery(rows, None, metric, 0) as g:
        idx, dist = g.search_top1(q, start, end)
        allv = g.range_distances(q, start, end)
    eidx, edist = oracle.top1_batch(rows, q, start, end, metric)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)
    for i in range(q.shape[0]):
        assert_bits_equal(allv[i], oracle.all_distances(rows, q[i], start, end, metric), f"range distances q{i}")


def test_end_zero_means_all_features(fir, oracle):
    rows = synth.make_gallery(8, 500, 96, L2)
    q, _ = synth.make_queries(8, rows, 4, L2)
    with fir.Gallery(rows, None, L2, 0) as g:
        a = g.search_top1(q, 0, 0)
        b = g.search_top1(q, 0, 96)
    assert np.array_equal(a[0], b[0]) and np.array_equal(bits(a[1]), bits(b[1]))


def test_ties_first_minimum_wins(fir, oracle):
    """F9: strict '<' in row order -- duplicates of the nearest row, across lanes, waves and tiles."""
    rows = synth.make_gallery(21, 5000, 64, L2)
    q, pick = synth.make_queries(21, rows, 6, L2)
    for i in range(6):
        src = rows[int(pick[i])].copy()
        q[i] = src
        for dup in (17 + i, 64 * 5 + 3, 64 * 40 + 63, 4999 - i):   # copies of the exact-match row, scattered
            rows[dup] = src
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, 64, L2)
    assert np.array_equal(idx, eidx)
    assert np.all(dist == 0)
    assert_bits_equal(dist, edist)


def test_near_ties_one_ulp(fir, oracle):
    rows = synth.make_gallery(22, 3000, 128, L2)
    q, _ = synth.make_queries(22, rows, 4, L2)
    base = rows[100].copy()
    for j, r in enumerate((200, 900, 1500, 2999)):
        v = base.copy()
        v[j] = np.nextafter(v[j], np.float32(2), dtype=np.float32)
        rows[r] = v
    q[:] = base
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, 128, L2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)


def test_not_found_sentinel_and_nan_rows(fir, oracle):
    """Nothing beats 100000 -> -1 (db_features.cpp:322-323); NaN rows never win (comparison false)."""
    rows = np.full((130, 8), 1.0e4, np.float32)          # mean squared distance to 0 is 1e8 > 100000
    q = np.zeros((3, 8), np.float32)
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_top1(q)
    assert np.all(idx == -1) and np.all(dist == np.float32(100000.0))
    for i in range(3):
        assert oracle.recognize_bf(rows, q[i], 0, 8, L2) == (-1, np.float32(100000.0))
    rows = synth.make_gallery(23, 300, 32, L2)
    rows[5, 3] = np.nan
    rows[77] = np.nan
    q, _ = synth.make_queries(23, rows, 4, L2)
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, 32, L2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)


def test_empty_gallery_and_empty_batch(fir):
    with fir.Gallery(np.zeros((0, 16), np.float32), None, L2, 0) as g:
        idx, dist = g.search_top1(np.ones((2, 16), np.float32))
        assert np.all(idx == -1) and np.all(dist == np.float32(100000.0))
    with fir.Gallery(np.ones((10, 16), np.float32), None, L2, 0) as g:
        idx, dist = g.search_top1(np.zeros((0, 16), np.float32))
        assert idx.size == 0


@pytest.mark.parametrize("seed,n,d", [(13, 257, 64), (17, 1000, 256), (101, 4099, 512)])
def test_chi2_top1_bit_exact(fir, oracle, seed, n, d):
    rows = synth.make_gallery(seed, n, d, CHI2)
    q, _ = synth.make_queries(seed, rows, 9, CHI2)
    rows[3] = 0.0   # a+b == 0 features are skipped (db_features.cpp:29)
    q[0, :10] = 0.0
    with fir.Gallery(rows, None, fir.METRIC_CHI2, 0) as g:
        idx, dist = g.search_top1(q)
    eidx, edist = oracle.top1_batch(rows, q, 0, d, CHI2)
    assert np.array_equal(idx, eidx)
    assert_bits_equal(dist, edist)


@pytest.mark.parametrize("seed,n,d", [(13, 257, 64), (17, 1000, 256), (101, 2000, 512)])
def test_kl_top1_within_tolerance(fir, oracle, seed, n, d):
    rows = synth.make_gallery(seed, n, d, KL)
    q, _ = synth.make_queries(seed, rows, 9, KL)
    with fir.Gallery(rows, None, fir.METRIC_KL, 0) as g:
        idx, dist = g.search_top1(q)
        allv = g.range_distances(q, 0, d)
    for i in range(q.shape[0]):
        ref = oracle.all_distances(rows, q[i], 0, d, KL)
        np.testing.assert_allclose(allv[i], ref, rtol=1e-5, atol=1e-9)
        order = np.argsort(ref, kind="stable")
        best, second = ref[order[0]], ref[order[1]]
        np.testing.assert_allclose(dist[i], best, rtol=1e-5, atol=1e-9)
        if second - best > 2e-5 * abs(best):
            assert idx[i] == order[0]


@pytest.mark.parametrize("metric", [L2, CHI2])
@pytest.mark.parametrize("k", [1, 3, 5, 8])
def test_topk_bit_exact(fir, oracle, metric, k):
    rows = synth.make_gallery(31, 6000, 256, metric)
    q, _ = synth.make_queries(31, rows, 7, metric)
    rows[4000] = rows[10]   # equal distances -> ascending row order
    rows[5999] = rows[10]
    with fir.Gallery(rows, None, metric, 0) as g:
        idx, dist = g.search_topk(q, k)
    for i in range(q.shape[0]):
        ei, ed = oracle.topk(rows, q[i], 0, 256, k, metric)
        assert np.array_equal(idx[i], ei), (i, idx[i], ei)
        assert_bits_equal(dist[i], ed)


def test_topk_fewer_rows_than_k(fir, oracle):
    rows = synth.make_gallery(32, 3, 16, L2)
    q, _ = synth.make_queries(32, rows, 2, L2)
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_topk(q, 5)
    for i in range(2):
        ei, ed = oracle.topk(rows, q[i], 0, 16, 5, L2)
        assert np.array_equal(idx[i], ei)
        assert_bits_equal(dist[i], ed)


def test_bruteforce_classifier_classes(fir, oracle):
    """BruteForceClassifier::recognize (ImageTesting.cpp:58-71): class of the best row, max_features prefix."""
    n, d = 3030, 1536
    rows = synth.make_gallery(13, n, d, L2)
    cls = synth.make_labels(n, 101)
    q, _ = synth.make_queries(13, rows, 6, L2)
    with fir.Gallery(rows, cls, L2, 0) as g:
        for maxf in (1536, 64, 256):
            idx, _ = g.search_top1(q, 0, maxf)
            got = g.classes_of(idx)
            exp = [oracle.bf_classifier(rows, cls, q[i], maxf, L2) for i in range(q.shape[0])]
            assert list(got) == exp
        assert list(g.classes_of(np.array([-1, 0, n - 1], np.int32))) == [-1, int(cls[0]), int(cls[n - 1])]


@pytest.mark.parametrize("metric", [L2, CHI2, KL])
def test_single_pair_feature_distance(fir, oracle, metric):
    a = synth.make_gallery(41, 1, 1536, metric)[0]
    b = synth.make_gallery(42, 1, 1536, metric)[0]
    for s, e in ((0, 1536), (0, 64), (64, 256), (7, 9)):
        got = fir.feature_distance(a, b, s, e, metric)
        exp = oracle.feature_distance(a, b, s, e, metric)
        if metric == KL:
            np.testing.assert_allclose(got, exp, rtol=1e-5)
        else:
            assert bits(got) == bits(exp)


def test_row_offset_and_shard_merge(fir, oracle):
    """Row-sharded galleries: min over the shards' packed keys == the unsharded answer."""
    rows = synth.make_gallery(51, 10000, 128, L2)
    q, pick = synth.make_queries(51, rows, 8, L2)
    rows[9000] = rows[123]
    q[1] = rows[123]     # tie across shards: the lower global row must win
    eidx, edist = oracle.top1_batch(rows, q, 0, 128, L2)
    for parts in (2, 3, 8):
        bounds = [(r * ((10000 + parts - 1) // parts), min(10000, (r + 1) * ((10000 + parts - 1) // parts))) for r in range(parts)]
        keys = []
        for lo, hi in bounds:
            with fir.Gallery(rows[lo:hi], None, L2, 0) as g:
                g.set_row_offset(lo)
                idx, dist = g.search_top1(q)
                keys.append(np.array([fir.key_pack(d_, i_) for d_, i_ in zip(dist, idx)], np.uint64))
        merged = np.minimum.reduce(keys)
        midx, mdist = fir.keys_unpack(merged)
        assert np.array_equal(midx, eidx)
        assert_bits_equal(mdist, edist)


def test_sharded_topk_merge_and_pnn_partial_sums(fir, oracle):
    """SURVEY 8e beyond top-1: the K nearest rows over row shards = K-way merge of the shards' packed keys; the PNN
    class scores = sum of the shards' partial scores (each over its training rows, divided by the global size)."""
    import torch

    import golden_cases as gc
    from fast_image_recognition_amd import sharding

    rows = synth.make_gallery(52, 9000, 96, CHI2)
    q, _ = synth.make_queries(52, rows, 7, CHI2)
    rows[8000] = rows[77]
    q[2] = rows[77]
    k = 5
    with fir.Gallery(rows, None, CHI2, 0) as g:
        widx, wdist = g.search_topk(q, k)
    for parts in (2, 3):
        per = (9000 + parts - 1) // parts
        stack = []
        for r in range(parts):
            lo, hi = r * per, min(9000, (r + 1) * per)
            with fir.Gallery(rows[lo:hi], None, CHI2, 0) as g:
                g.set_row_offset(lo)
                idx, dist = g.search_topk(q, k)
            keys = np.array([[fir.key_pack(dist[i, j], idx[i, j]) for j in range(k)] for i in range(len(q))], np.uint64)
            stack.append(sharding.keys_as_int64(torch.from_numpy(keys.view(np.int64))))
        merged = sharding.keys_from_int64(sharding.merge_topk_keys(torch.stack(stack), k)).numpy().view(np.uint64)
        midx, mdist = fir.keys_unpack(merged.reshape(-1))
        assert np.array_equal(midx.reshape(-1, k), widx)
        assert_bits_equal(mdist.reshape(-1, k), wdist)

    x, lab, ncls = gc.cls_case(seed=53, n=700, d=64, n_classes=9)
    order = np.argsort(lab, kind="stable")
    tr, tcls, qs = x[order][:600], lab[order][:600], x[order][600:640]
    avg = tr.mean(0)
    whole = fir.ClsModel(tr, tcls, ncls, avg, 0)
    wbest, wscores = whole.pnn_predict(qs)
    whole.close()
    total = np.zeros_like(wscores)
    for lo, hi in ((0, 250), (250, 251), (251, 600)):
        m = fir.ClsModel(tr[lo:hi], tcls[lo:hi], ncls, avg, 0)
        m.set_total_training_size(600)
        total += m.pnn_predict(qs)[1]
        m.close()
    assert np.allclose(total, wscores, rtol=1e-12, atol=1e-300)
    assert np.array_equal(sharding.first_max_class(torch.from_numpy(total)).numpy(), wbest)


def test_topk_candidate_lists_on_large_galleries(fir, oracle):
    """Batches of >= 8 queries over >= 65536 rows take the candidate-list form of the top-K scan (threshold from a row
    sample, append scan, K smallest of each list). Same answers as the oracle -- also with ties at the threshold, with
    more ties than a list holds (falls back to the register-list scan) and with rows that never qualify."""
    n, d, k = 70000, 64, 5
    rows = synth.make_gallery(71, n, d, L2)
    q, _ = synth.make_queries(71, rows, 11, L2)
    rows[100:106] = rows[69990]                  # seven equal rows: ties inside and at the K-th place, ordered by row
    q[0] = rows[69990]
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_topk(q, k)
        idx8, dist8 = g.search_topk(q, 8, 4, 60)       # another K and a sub-range
    for j in range(len(q)):
        ei, ed = oracle.topk(rows, q[j], 0, d, k, L2)
        assert np.array_equal(idx[j], ei), j
        assert_bits_equal(dist[j], ed)
        ei, ed = oracle.topk(rows, q[j], 4, 60, 8, L2)
        assert np.array_equal(idx8[j], ei), j
        assert_bits_equal(dist8[j], ed)
    assert list(idx[0]) == [100, 101, 102, 103, 104]
    rows[1000:9000] = rows[69990]                # 8000 more copies: the list overflows, the fallback answers
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_topk(q, k)
    for j in range(len(q)):
        ei, ed = oracle.topk(rows, q[j], 0, d, k, L2)
        assert np.array_equal(idx[j], ei), j
        assert_bits_equal(dist[j], ed)
    big_q = np.tile(q, (120, 1))[:1100]          # more queries than one candidate-list batch (1024)
    with fir.Gallery(rows, None, L2, 0) as g:
        bidx, bdist = g.search_topk(big_q, k)
    assert np.array_equal(bidx[:11], idx) and np.array_equal(bidx[1089:1100], idx[:11]) and np.array_equal(bits(bdist[:11]), bits(dist))
    far = np.full_like(q, 3000.0)                # nothing within 100000: every slot stays -1 / 100000
    with fir.Gallery(rows, None, L2, 0) as g:
        idx, dist = g.search_topk(far, k)
    assert (idx == -1).all() and (dist == np.float32(100000.0)).all()


@pytest.mark.parametrize("scale", [1e-18, 1e-20, 3e-23])
def test_denormal_products_and_sums_match_the_cpu(fir, oracle, scale):
    """Un-normalised inputs of tiny magnitude: (q - g)^2 and the running sums fall into the float denormal range (and
    partly to zero). The GPU keeps denormals like the reference's SSE arithmetic: same bits, same first minimum."""
    rows = (synth.make_gallery(91, 3000, 96, L2) * np.float32(scale)).astype(np.float32)
    q = (synth.make_queries(91, rows, 9, L2)[0] * np.float32(scale)).astype(np.float32)
    for metric in (L2, CHI2):
        with fir.Gallery(rows, None, metric, 0) as g:
            idx, dist = g.search_top1(q)
            allq = g.range_distances(q[:2])
            k_idx, k_dist = g.search_topk(q, 3)
        for j in range(len(q)):
            ei, ed = oracle.recognize_bf(rows, q[j], 0, 96, metric)
            assert idx[j] == ei and bits(dist[j]) == bits(ed), (metric, j, dist[j], ed)
            ki, kd = oracle.topk(rows, q[j], 0, 96, 3, metric)
            assert np.array_equal(k_idx[j], ki) and np.array_equal(bits(k_dist[j]), bits(kd))
        for j in range(2):
            assert np.array_equal(bits(allq[j]), bits(oracle.all_distances(rows, q[j], 0, 96, metric)))
    # the hand-scheduled packed-f32 kernels (8 and 16 queries per pass) on a gallery large enough to select them
    big = (synth.make_gallery(92, 70000, 64, L2) * np.float32(scale)).astype(np.float32)
    qb16 = (synth.make_queries(92, big, 16, L2)[0] * np.float32(scale)).astype(np.float32)
    with fir.Gallery(big, None, L2, 0) as g:
        for qpp in (8, 16):
            g.set_tuning(qpp, 0)
            idx, dist = g.search_top1(qb16)
            for j in range(16):
                ei, ed = oracle.recognize_bf(big, qb16[j], 0, 64, L2)
                assert idx[j] == ei and bits(dist[j]) == bits(ed), (qpp, j, dist[j], ed)


def test_mixed_sign_inputs_follow_the_reference_guards(fir, oracle):
    """Negative and zero features: chi-square / KL only add a feature when lhs + rhs > 0, KL only the terms whose own
    value is > 0 (db_features.cpp:29-36). Same rows, same bits (chi-square), as the CPU."""
    rng = np.random.default_rng(7)
    rows = (rng.random((5000, 40), dtype=np.float32) - np.float32(0.35)).astype(np.float32)
    rows[rng.random(rows.shape) < 0.1] = 0
    q = (rng.random((9, 40), dtype=np.float32) - np.float32(0.35)).astype(np.float32)
    q[0] = -rows[17]                              # lhs + rhs == 0 everywhere: nothing is added, distance 0
    for metric in (CHI2, KL):
        with fir.Gallery(rows, None, metric, 0) as g:
            idx, dist = g.search_top1(q)
            allq = g.range_distances(q[:3], 3, 37)
        for j in range(len(q)):
            ei, ed = oracle.recognize_bf(rows, q[j], 0, 40, metric)
            if metric == CHI2:
                assert idx[j] == ei and bits(dist[j]) == bits(ed), (j, dist[j], ed)
            else:
                assert abs(float(dist[j]) - float(ed)) <= 1e-5 * abs(float(ed)) + 1e-7, (j, dist[j], ed)
        for j in range(3):
            exp = oracle.all_distances(rows, q[j], 3, 37, metric)
            if metric == CHI2:
                assert np.array_equal(bits(allq[j]), bits(exp))
            else:
                assert np.allclose(allq[j], exp, rtol=1e-5, atol=1e-7)


def test_handles_release_their_device_memory(fir):
    """Create / use / destroy every kind of handle repeatedly: free device memory ends where it started."""
    import torch

    import golden_cases as gc

    rows = synth.make_gallery(81, 20000, 256, L2)
    cls = synth.make_labels(20000, 40)
    q, _ = synth.make_queries(81, rows, 16, L2)
    x, lab, ncls = gc.cls_case(seed=82, n=600, d=64, n_classes=6)
    order = np.argsort(lab, kind="stable")

    def cycle():
        with fir.Gallery(rows, cls, L2, 0) as g:
            g.search_top1(q)
            g.search_topk(q, 5)
            g.twd_conventional(q, 40, 0, 0.24, 64)
            g.twd_proposed(q, 32, 0.7)
            g.rows_distances(q[:2], np.arange(10, dtype=np.int32).reshape(2, 5))
            dem = fir.Dem(g, 3, 6)
            dem.likelihoods(q[:3])
            dem.close()
            with fir.GemmSearch(g) as m:
                keys = torch.empty(16, dtype=torch.int64, device="cuda")
                tq = torch.from_numpy(q).cuda()
                m.search_top1_keys_dev(tq.data_ptr(), 16, keys.data_ptr())
                torch.cuda.synchronize()
        c = fir.ClsModel(x[order], lab[order], ncls, x.mean(0), 0)
        c.pnn_predict(x[:4])
        c.close()
        f = fir.Fpnn(x[order], lab[order], ncls, x.mean(0), x.std(0), 1.0, 0)
        f.predict(x[:4])
        f.close()

    cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(5):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 64 << 20, (free0, free1)       # allocator granularity, not a per-cycle leak (one cycle allocates > 200 MB)


def test_errors_are_reported_not_crashed(fir):
    rows = synth.make_gallery(61, 100, 32, L2)
    with fir.Gallery(rows, None, L2, 0) as g:
        with pytest.raises(fir.FirError):
            g.search_top1(np.zeros((1, 32), np.float32), 10, 5)
        with pytest.raises(fir.FirError):
            g.search_top1(np.zeros((1, 32), np.float32), 0, 33)
        with pytest.raises(fir.FirError):
            g.search_topk(np.zeros((1, 32), np.float32), 99)
        with pytest.raises(fir.FirError):
            g.classes_of(np.zeros(1, np.int32))    # created without labels
    with pytest.raises(fir.FirError):
        fir.Gallery(rows, None, L2, 99)
