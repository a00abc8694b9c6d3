"""Chi-square / KL scans pick their f32 division sequence from the operands' range (include/fir_amd.h,
fir_gallery_value_range; csrc/fir_common.h). Both sequences are the same arithmetic, so results must not depend on
which one ran: the same queries give the same bits whether or not an out-of-range query rides in the batch."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu


def _l1_rows(seed, n, d, lo_exp=None):
    rows = synth.make_gallery(seed, n, d, 1)          # non-negative, |x| < 1e-4 -> 0, L1-normalised (db_features.cpp:85-101)
    if lo_exp is not None:
        rows = rows.copy()
        rows[::7, 3] = np.float32(2.0 ** lo_exp)      # plant values at the low edge of the plain range
    return rows


@pytest.mark.parametrize("metric", [1, 2])
@pytest.mark.parametrize("n,d,qb", [(4099, 512, 13), (1000, 1536, 8), (257, 130, 3), (20000, 64, 33)])
def test_results_do_not_depend_on_the_division_sequence(fir, oracle, metric, n, d, qb):
    rows = _l1_rows(31 + n, n, d, lo_exp=-26)
    queries, _ = synth.make_queries(31 + n, rows, qb, 1)
    poisoned = np.vstack([queries, -np.abs(queries[:1])])          # a negative query value: outside the plain range
    with fir.Gallery(rows, None, metric, 0) as g:
        idx, dist = g.search_top1(queries)
        assert g.value_range() == (True, True)
        idx2, dist2 = g.search_top1(poisoned)
        assert g.value_range() == (True, False)
        kidx, kdist = g.search_topk(queries, 5)
        assert g.value_range() == (True, True)
        kidx2, kdist2 = g.search_topk(poisoned, 5)
        m = min(qb, 7)
        all1 = g.range_distances(queries[:m], 0, d)
        all2 = g.range_distances(np.vstack([queries[:m], poisoned[-1:]]), 0, d)
    assert np.array_equal(idx, idx2[:qb])
    assert np.array_equal(dist.view(np.uint32), dist2[:qb].view(np.uint32))
    assert np.array_equal(kidx, kidx2[:qb])
    assert np.array_equal(kdist.view(np.uint32), kdist2[:qb].view(np.uint32))
    assert np.array_equal(all1.view(np.uint32), all2[:m].view(np.uint32))
    if metric == 1:                                                 # chi-square is bit-exact against the reference arithmetic
        eidx, edist = oracle.top1_batch(rows, queries, 0, d, 1)
        assert np.array_equal(idx, eidx)
        assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))


@pytest.mark.parametrize("metric", [1, 2])
def test_gallery_values_outside_the_range_keep_the_full_sequence(fir, oracle, metric):
    n, d, qb = 3000, 256, 9
    rows = _l1_rows(77, n, d)
    queries, _ = synth.make_queries(77, rows, qb, 1)
    odd = rows.copy()
    odd[1234, 17] = np.float32(1e-41)                              # a denormal: v_div_scale would scale this one
    with fir.Gallery(odd, None, metric, 0) as g:
        idx, dist = g.search_top1(queries)
        assert g.value_range() == (False, True)
    eidx, edist = oracle.top1_batch(odd, queries, 0, d, metric)
    assert np.array_equal(idx, eidx)
    if metric == 1:
        assert np.array_equal(dist.view(np.uint32), edist.view(np.uint32))
    else:
        assert np.allclose(dist, edist, rtol=1e-5, atol=0)          # KL: the tolerance north_star states


def test_l2_galleries_report_the_range_too(fir):
    rows = synth.make_gallery(5, 500, 128, 0)
    q, _ = synth.make_queries(5, rows, 4, 0)
    with fir.Gallery(rows, None, 0, 0) as g:
        g.search_top1(q)
        assert g.value_range() == (True, True)


@pytest.mark.parametrize("form", ["2", "1"])
def test_chi2_nomination_scan_returns_the_exact_scans_keys(fir, oracle, form, monkeypatch):
    """form 2 (default): the harmonic form chi2 = sum(l) + sum(r) - 4 sum 1/(1/l_k + 1/r_k), one reciprocal per element and query, threshold
    widened ADDITIVELY by its error bound; form 1: (l - r)^2 * rcp(l + r), threshold widened relatively (FIR_CHI2_NOMINATION).
    Chi-square batches over a large plain-range gallery take a nomination scan (1-ulp reciprocal) + exact re-rank by default
    (fir_capi.hip: topk_lists_dev). Keys must be the exact scan's bit for bit -- top-1 and top-5, whole range and a sub-range,
    with exact duplicates (ties by row) and near-ties one ulp apart -- and the oracle's on a sample. A negative value in a
    query leaves the plain range: the exact path answers that call."""
    import synth

    monkeypatch.setenv("FIR_CHI2_NOMINATION", form)
    n, d, qb = 70000, 128, 40
    rows = synth.make_gallery(61, n, d, 1)
    q, _ = synth.make_queries(61, rows, qb, 1)
    rows[n - 3] = rows[17]
    q[2] = rows[17]                                   # exact tie (distance 0): first row wins
    rows[5000] = rows[4000]
    rows[5000, 7] = np.nextafter(rows[5000, 7], np.float32(1))     # near-tie one ulp apart
    q[3] = 0.5 * (rows[4000] + rows[123])
    with fir.Gallery(rows, None, fir.METRIC_CHI2, 0) as g:
        a1 = g.search_top1(q)                         # default dispatch: nomination
        a5 = g.search_topk(q, 5)
        s1 = g.search_top1(q, 32, 96)
        g.set_tuning(queries_per_pass=8)              # a pinned tile size keeps top-1 on the exact scan
        e1 = g.search_top1(q)
        es1 = g.search_top1(q, 32, 96)
        g.set_tuning(queries_per_pass=-1)
        qn = q.copy()
        qn[0, 0] = -1e-3                               # outside the plain range
        n1 = g.search_top1(qn)
    assert np.array_equal(a1[0], e1[0]) and np.array_equal(a1[1].view(np.uint32), e1[1].view(np.uint32))
    assert np.array_equal(s1[0], es1[0]) and np.array_equal(s1[1].view(np.uint32), es1[1].view(np.uint32))
    assert np.array_equal(a5[0][:, 0], a1[0]) and a1[0][2] == 17
    for i in (0, 2, 3, 11, 39):
        ei, ed = oracle.topk(rows, q[i], 0, d, 5, 1)
        assert np.array_equal(a5[0][i], ei) and np.array_equal(a5[1][i].view(np.uint32), ed.view(np.uint32))
    oi, od = oracle.recognize_bf(rows, qn[0], 0, d, 1)
    assert n1[0][0] == oi and np.float32(n1[1][0]).view(np.uint32) == np.float32(od).view(np.uint32)


def test_kl_nomination_scan_returns_the_exact_scans_keys(fir, oracle, monkeypatch):
    """KL batches over a large plain-range gallery: the nomination scan adds up (l + r) log2 (l + r) only (kKLEnt: the other two
    sums of KL = ln2 (sum(l log2 l + l) + sum(r log2 r + r) - sum (l + r) log2 (l + r)) belong to the query and to the row), the
    threshold is widened by the form's error bound and the appended rows are re-ranked with the exact scan's arithmetic
    (fir_capi.hip: topk_lists_dev). Keys must be the exact scan's bit for bit -- top-1 and top-5, whole range and a sub-range, with
    exact duplicates (ties by row), zeros on both sides and a near-tie -- and the oracle's within the KL tolerance on a sample."""
    import synth

    n, d, qb = 70000, 128, 40
    rows = synth.make_gallery(71, n, d, 2)
    q, _ = synth.make_queries(71, rows, qb, 2)
    rows[:, 5] = 0                                    # zeros on both sides: the term is skipped by the reference
    q[:, 5] = 0
    rows[1000:1100, 9] = 0
    q[7, 11] = 0
    rows[n - 3] = rows[17]
    q[2] = rows[17]                                   # exact tie (distance 0): first row wins
    rows[5000] = rows[4000]
    rows[5000, 7] = np.nextafter(rows[5000, 7], np.float32(1))     # near-tie one ulp apart
    q[3] = 0.5 * (rows[4000] + rows[123])
    # a query that is a one-ulp neighbour of two gallery rows -- its KL value l log(2l/s) + r log(2r/s) cancels to a tiny number
    # that can round below zero, and so can a sampled threshold: it must be widened like any other (the "no threshold" sentinel is -inf)
    rows[6000] = rows[6100]
    rows[6000, 3] = np.nextafter(rows[6000, 3], np.float32(0))
    q[4] = rows[6100]
    q[4, 3] = np.nextafter(q[4, 3], np.float32(1))
    with fir.Gallery(rows, None, fir.METRIC_KL, 0) as g:
        a1 = g.search_top1(q)                         # default dispatch: nomination
        a5 = g.search_topk(q, 5)
        s1 = g.search_top1(q, 32, 96)
        monkeypatch.setenv("FIR_NO_CHI2_NOMINATION", "1")
        e1 = g.search_top1(q)
        e5 = g.search_topk(q, 5)
        es1 = g.search_top1(q, 32, 96)
        monkeypatch.delenv("FIR_NO_CHI2_NOMINATION")
    assert np.array_equal(a1[0], e1[0]) and np.array_equal(a1[1].view(np.uint32), e1[1].view(np.uint32))
    assert np.array_equal(s1[0], es1[0]) and np.array_equal(s1[1].view(np.uint32), es1[1].view(np.uint32))
    assert np.array_equal(a5[0], e5[0]) and np.array_equal(a5[1].view(np.uint32), e5[1].view(np.uint32))
    assert np.array_equal(a5[0][:, 0], a1[0]) and a1[0][2] == 17
    # one handle, the metric switched between calls: the per-row constants of the two nomination forms (row sums / row entropies)
    # are rebuilt, not reused
    with fir.Gallery(rows, None, fir.METRIC_CHI2, 0) as g:
        c1 = g.search_top1(q)
        g.set_metric(fir.METRIC_KL)
        k1 = g.search_top1(q)
        g.set_metric(fir.METRIC_CHI2)
        c2 = g.search_top1(q)
    assert np.array_equal(k1[0], e1[0]) and np.array_equal(k1[1].view(np.uint32), e1[1].view(np.uint32))
    assert np.array_equal(c1[0], c2[0]) and np.array_equal(c1[1].view(np.uint32), c2[1].view(np.uint32))
    for i in (0, 2, 7, 11, 39):
        ei, ed = oracle.topk(rows, q[i], 0, d, 5, 2)
        assert np.allclose(a5[1][i], ed, rtol=1e-5, atol=1e-9)          # device log vs glibc logf: the KL tolerance of test_gpu_golden
        gaps = np.diff(ed) > 4e-5 * np.abs(ed[1:])
        if gaps.all():
            assert np.array_equal(a5[0][i], ei)


@pytest.mark.parametrize("metric", [1, 2])
def test_nomination_scans_over_ranges_with_ragged_edges(fir, metric, monkeypatch):
    """The chi-square (harmonic, two terms per reciprocal) and KL (entropy form) nomination scans over feature ranges whose ends are not
    multiples of 4 (the edge features go through the one-term forms), a range inside one 4-feature chunk, and batches that fill one
    tile of 8 queries, one and a half, and an odd number of tiles: the exact scan's keys, bit for bit."""
    import synth

    n, d = 66000, 96
    rows = synth.make_gallery(83, n, d, metric)
    q, _ = synth.make_queries(83, rows, 40, metric)
    rows[:, 7] = 0
    q[::3, 8] = 0
    with fir.Gallery(rows, None, metric, 0) as g:
        for (a, b) in ((5, 93), (0, 96), (33, 35), (3, 70)):
            for nq in (8, 12, 40):
                got1 = g.search_top1(q[:nq], a, b)
                got5 = g.search_topk(q[:nq], 5, a, b)
                monkeypatch.setenv("FIR_NO_CHI2_NOMINATION", "1")
                exp1 = g.search_top1(q[:nq], a, b)
                exp5 = g.search_topk(q[:nq], 5, a, b)
                monkeypatch.delenv("FIR_NO_CHI2_NOMINATION")
                assert np.array_equal(got1[0], exp1[0]) and np.array_equal(got1[1].view(np.uint32), exp1[1].view(np.uint32)), (a, b, nq)
                assert np.array_equal(got5[0], exp5[0]) and np.array_equal(got5[1].view(np.uint32), exp5[1].view(np.uint32)), (a, b, nq)


@pytest.mark.parametrize("metric", [1, 2])
def test_thresholds_from_nomination_metric_samples_lose_no_row(fir, metric, monkeypatch):
    """Since round 4 the row samples that give the append scan its threshold run the nomination metric themselves (fir_capi.hip:
    topk_lists_dev, approx_samples): a sampled minimum is only within B of the reference's value, so the threshold is widened by
    2.5 B. The tight case: the query's nearest rows lie INSIDE the sampled rows (the first 16 384), so the threshold is as low as
    it can get, and rows outside the sample tie with them exactly or sit one ulp on either side -- every one of them has to be
    appended, and the ties have to resolve by row as the exact scan resolves them. FIR_EXACT_SAMPLES=1 (the exact metric's
    samples, 1.5 B) must give the same keys."""
    import synth

    n, d, qb = 70000, 128, 24
    rows = synth.make_gallery(97, n, d, metric)
    q = np.empty((qb, d), np.float32)
    for i in range(qb):
        src = 40 * i + 3                                           # inside the sample of every group (tile t belongs to group t mod K)
        q[i] = rows[src] * (1 + 0.01 * np.cos(np.arange(d) + i)).astype(np.float32)
        q[i] /= q[i].sum()
        rows[20000 + 7 * i] = rows[src]                            # an exact copy outside the sample: the tie goes to the earlier row
        up = rows[src].copy()
        up[i % d] = np.nextafter(up[i % d], np.float32(1))
        rows[40000 + 11 * i] = up                                  # one ulp away in one feature, outside the sample
        dn = rows[src].copy()
        dn[(i + 1) % d] = np.nextafter(dn[(i + 1) % d], np.float32(0))
        rows[60000 + 13 * i] = dn
    with fir.Gallery(rows, None, metric, 0) as g:
        a1 = g.search_top1(q)
        a5 = g.search_topk(q, 5)
        assert "k_nominate" in g.last_dispatch()["kernel"]
        monkeypatch.setenv("FIR_EXACT_SAMPLES", "1")
        x1 = g.search_top1(q)
        x5 = g.search_topk(q, 5)
        monkeypatch.delenv("FIR_EXACT_SAMPLES")
        monkeypatch.setenv("FIR_NO_CHI2_NOMINATION", "1")
        e1 = g.search_top1(q)
        e5 = g.search_topk(q, 5)
        monkeypatch.delenv("FIR_NO_CHI2_NOMINATION")
    for got1, got5 in ((a1, a5), (x1, x5)):
        assert np.array_equal(got1[0], e1[0]) and np.array_equal(got1[1].view(np.uint32), e1[1].view(np.uint32))
        assert np.array_equal(got5[0], e5[0]) and np.array_equal(got5[1].view(np.uint32), e5[1].view(np.uint32))
    # the four planted rows of every query are its four nearest, whatever their order
    for i in range(qb):
        assert {40 * i + 3, 20000 + 7 * i, 40000 + 11 * i, 60000 + 13 * i} <= set(int(v) for v in e5[0][i])
