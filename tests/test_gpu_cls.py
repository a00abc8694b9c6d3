"""GPU parity of the double-precision kNN / PNN path (qt_cpp/classification.cpp:116-226):
distance sums bit-identical to the oracle; predicted classes identical to the oracle and to the
REAL reference's predictions recorded in tests/golden; PNN class scores within 1e-12 relative."""
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))


def test_reference_predictions_reproduced(fir, oracle):
    x, lab, ncls = gc.cls_case()
    train, tcls, test = GOLD["cls/train"], GOLD["cls/train_class"], GOLD["cls/test"]
    tr, avg = x[train], GOLD["cls/avg"]
    q = x[test]
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        sums = m.distance_sums(q)
        knn1, knn3 = m.knn_predict(q, 1), m.knn_predict(q, 3)
        pnn, scores = m.pnn_predict(q)
        seq, _ = m.pnn_predict_seq(q)
    assert np.array_equal(seq, GOLD["cls/pnn_seq"])
    assert np.array_equal(knn1, GOLD["cls/knn1"])
    assert np.array_equal(knn3, GOLD["cls/knn3"])
    assert np.array_equal(pnn, GOLD["cls/pnn"])
    d = tr.shape[1]
    for i in (0, 7, len(test) - 1):
        _, dist = oracle.knn_predict(tr, tcls, avg, ncls, q[i], 1)          # mean distances (sum / d)
        assert np.array_equal((sums[i] / d).view(np.uint64), dist.view(np.uint64))
        _, es = oracle.pnn_predict(tr, tcls, avg, ncls, q[i])
        np.testing.assert_allclose(scores[i], es, rtol=1e-12, atol=0)


@pytest.mark.parametrize("seed,n,d,ncls,frac", [(3, 500, 64, 10, 0.5), (4, 1200, 257, 25, 0.4), (5, 200, 3, 4, 0.7), (6, 130, 2100, 5, 0.5),
                                                  (7, 135, 1, 3, 0.5), (8, 129, 33, 2, 0.5), (9, 6000, 31, 40, 0.9), (10, 20000, 96, 101, 0.95),
                                                  (11, 300, 640, 7, 0.43)])
def test_matches_oracle_on_fresh_data(fir, oracle, seed, n, d, ncls, frac):
    x, lab, _ = gc.cls_case(seed=seed, n=n, d=d, n_classes=ncls)
    rng = np.random.default_rng(seed)
    is_train = rng.random(n) < frac
    order = np.argsort(lab[is_train], kind="stable")
    tr, tcls = x[is_train][order], lab[is_train][order]
    q = x[~is_train][:23]
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        sums = m.distance_sums(q)
        knn = {k: m.knn_predict(q, k) for k in (1, 3, 5)}
        pnn, scores = m.pnn_predict(q)
        seq, seq_chunks = m.pnn_predict_seq(q)
    for i in range(q.shape[0]):
        es_, ec_ = oracle.pnn_predict_seq(tr, tcls, avg, ncls, q[i])
        assert (seq[i], seq_chunks[i]) == (es_, ec_)
        e1, dist = oracle.knn_predict(tr, tcls, avg, ncls, q[i], 1)
        assert np.array_equal((sums[i] / d).view(np.uint64), dist.view(np.uint64))
        assert knn[1][i] == e1
        assert knn[3][i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 3)[0]
        assert knn[5][i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 5)[0]
        ep, es = oracle.pnn_predict(tr, tcls, avg, ncls, q[i])       # d > 2000 exercises var/10 (classification.cpp:192-193)
        np.testing.assert_allclose(scores[i], es, rtol=1e-12, atol=1e-300)
        if np.sort(es)[-1] > 0 and np.sort(es)[-1] > np.sort(es)[-2] * (1 + 1e-9):
            assert pnn[i] == ep


def test_batches_larger_than_the_internal_distance_table(fir, oracle):
    """200 000 training rows: the qb x nt distance table is capped at 1 GiB, so 700 queries are answered in two internal
    batches -- same answers as asking for them in pieces, and as the oracle on a sample."""
    rng = np.random.default_rng(12)
    nt, d, ncls = 200_000, 4, 5
    x = rng.random((nt, d))
    lab = np.sort(rng.integers(0, ncls, nt)).astype(np.int32)
    x += lab[:, None] * 0.15
    q = rng.random((700, d)) + rng.integers(0, ncls, 700)[:, None] * 0.15
    avg = x.mean(0)
    with fir.ClsModel(x, lab, ncls, avg, 0) as m:
        knn = m.knn_predict(q, 3)
        pnn, _ = m.pnn_predict(q)
        knn_parts = np.concatenate([m.knn_predict(q[:100], 3), m.knn_predict(q[100:], 3)])
        pnn_parts = np.concatenate([m.pnn_predict(q[:300])[0], m.pnn_predict(q[300:])[0]])
    assert np.array_equal(knn, knn_parts) and np.array_equal(pnn, pnn_parts)
    for i in (0, 1, 350, 667, 668, 669, 699):
        assert knn[i] == oracle.knn_predict(x, lab, avg, ncls, q[i], 3)[0], i
    assert np.mean(knn == pnn) > 0.5


def test_classes_smaller_than_k_and_bad_arguments(fir, oracle):
    x, lab, ncls = gc.cls_case(seed=9, n=40, d=16, n_classes=8)        # 5 rows per class, k = 8 never reached
    order = np.argsort(lab, kind="stable")
    tr, tcls = x[order][:-3], lab[order][:-3]                          # last class has 2 rows
    _, _, avg, _ = oracle.train_stats(tr)
    q = x[:6]
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        got = m.knn_predict(q, 8)
        for i in range(6):
            assert got[i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 8)[0]
        with pytest.raises(fir.FirError):
            m.knn_predict(q, 9)
    with pytest.raises(fir.FirError):
        fir.ClsModel(tr, tcls[::-1].copy(), ncls, avg, 0)               # classes must be non-decreasing


@pytest.mark.parametrize("shards", [2, 3])
def test_knn_vote_over_training_row_shards(fir, oracle, shards):
    """SURVEY 8e for the kNN classifier: every shard of the training rows exports its k nearest mean distances per class
    (fir_cls_knn_class_nearest); the k-th of the merged lists decides (sharding.merge_knn_class_nearest / knn_class_of).
    Same classes as the unsharded call and as the oracle, including k larger than every class (no winner: largest class)."""
    import torch
    from fast_image_recognition_amd import sharding

    x, lab, ncls = gc.cls_case(seed=21, n=600, d=48, n_classes=9)
    order = np.argsort(lab, kind="stable")
    tr, tcls = x[order][:520], lab[order][:520]
    q = x[order][520:560]
    _, _, avg, _ = oracle.train_stats(tr)
    sizes = np.bincount(tcls, minlength=ncls)
    for k in (1, 3, 8):
        if k == 8:                                                     # shrink every class below k
            keep = np.concatenate([np.flatnonzero(tcls == c)[:5 + (c == 4)] for c in range(ncls)])
            tr_k, tcls_k = tr[keep], tcls[keep]
        else:
            tr_k, tcls_k = tr, tcls
        sizes = np.bincount(tcls_k, minlength=ncls)
        parts = []
        for r in range(shards):
            lo, hi = sharding.shard_bounds(tr_k.shape[0], shards, r)
            with fir.ClsModel(tr_k[lo:hi], tcls_k[lo:hi], ncls, avg, 0) as m:
                near = m.knn_class_nearest(q, k)
            assert near.shape == (q.shape[0], ncls, k)
            assert np.all(np.diff(near, axis=2) >= 0)
            parts.append(torch.from_numpy(near))
        kth = sharding.merge_knn_class_nearest(torch.stack(parts), k)
        got = sharding.knn_class_of(kth, sizes).numpy()
        with fir.ClsModel(tr_k, tcls_k, ncls, avg, 0) as m:
            whole = m.knn_predict(q, k)
        assert np.array_equal(got, whole), k
        for i in range(q.shape[0]):
            assert got[i] == oracle.knn_predict(tr_k, tcls_k, avg, ncls, q[i], k)[0], (k, i)


def _big_training_set(n, d, ncls, seed):
    """Class-major float64 training rows: class centres + noise, so that the PNN scores of the right class are not denormal."""
    rng = np.random.default_rng(seed)
    centres = rng.random((ncls, d))
    tcls = np.sort(rng.integers(0, ncls, n)).astype(np.int32)
    tr = centres[tcls] + 0.004 * rng.standard_normal((n, d))
    q = centres[rng.integers(0, ncls, 24)] + 0.004 * rng.standard_normal((24, d))
    return tr, tcls, q


def test_k3_pnn_and_knn_at_120k_rows_against_the_oracle(fir, oracle):
    """K3 at scale: 120 000 x 512 float64 training rows, PNN class scores, PNN / kNN classes and the
    distance sums of sampled queries against the oracle (one query ~0.15 s on the CPU)."""
    n, d, ncls = 120_000, 512, 300
    tr, tcls, q = _big_training_set(n, d, ncls, 77)
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        pnn, scores = m.pnn_predict(q)
        knn1, knn3 = m.knn_predict(q, 1), m.knn_predict(q, 3)
        sums = m.distance_sums(q[:2])
    for i in (0, 5, 11, 23):
        ec, es = oracle.pnn_predict(tr, tcls, avg, ncls, q[i])
        assert pnn[i] == ec
        np.testing.assert_allclose(scores[i], es, rtol=1e-11, atol=1e-300)     # ~400 rows per class summed in another order
        e1, dist = oracle.knn_predict(tr, tcls, avg, ncls, q[i], 1)
        assert knn1[i] == e1 and knn3[i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 3)[0]
        if i < 2:
            assert np.array_equal((sums[i] / d).view(np.uint64), dist.view(np.uint64))      # per-row sums: bit-identical
    assert scores.max() > 1e-200                                              # the test is not comparing zeros


def test_k3_distance_sums_are_the_same_bits_whatever_the_tiling(fir, oracle, monkeypatch):
    """The scan over a training set streamed from HBM takes two tiles of eight queries per read of the rows (k_cls_scan_lds<8, 2>),
    one when the call has a single tile or FIR_CLS_ONE_TILE is set: the per-row sums are the same bits in every case -- 1, 8, 9
    (a ragged second tile), 24 (an odd number of tiles) and 37 queries -- and the oracle's on sampled rows."""
    n, d, ncls = 70_000, 512, 100
    tr, tcls, q = _big_training_set(n, d, ncls, 91)
    q = np.concatenate([q, q[:13] * 0.5 + tr[:13] * 0.5])[:37]
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        full = m.distance_sums(q)
        for nq in (1, 8, 9, 24):
            part = m.distance_sums(q[:nq])
            assert np.array_equal(part.view(np.uint64), full[:nq].view(np.uint64)), nq
    monkeypatch.setenv("FIR_CLS_ONE_TILE", "1")
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        one = m.distance_sums(q)
    assert np.array_equal(one.view(np.uint64), full.view(np.uint64))
    _, dist = oracle.knn_predict(tr, tcls, avg, ncls, q[36], 1)
    assert np.array_equal((full[36] / d).view(np.uint64), dist.view(np.uint64))


def test_pnn_over_training_row_shards_inside_the_library(fir, oracle):
    """fir_cls_create_sharded / fir_cls_sharded_pnn_predict: partial class sums over the GLOBAL training-set size, added on the
    device and by ncclAllReduce(ncclSum, ncclDouble). Same classes as the one-handle call, scores within 1e-12 (the order
    of the additions is all that differs); a near-tie of two classes is graded by that tolerance, not by the arg-max."""
    n, d, ncls = 9000, 96, 31
    tr, tcls, q = _big_training_set(n, d, ncls, 5)
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        b0, s0 = m.pnn_predict(q)
    for spd in (1, 3, 8):
        with fir.ShardedClsModel(tr, tcls, ncls, avg, devices=[0], shards_per_device=spd) as s:
            b, sc = s.pnn_predict(q)
        np.testing.assert_allclose(sc, s0, rtol=1e-12, atol=1e-300)
        top2 = np.sort(s0, axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-11 * top2[:, 1]
        assert np.array_equal(b[clear], b0[clear]) and clear.sum() >= 20
    ec, es = oracle.pnn_predict(tr, tcls, avg, ncls, q[3])
    assert b0[3] == ec
    np.testing.assert_allclose(s0[3], es, rtol=1e-12, atol=1e-300)


def test_large_query_batch_against_a_tiny_training_set(fir, oracle):
    """More than 65 535 queries in one call against a training set small enough that the distance table allows it: the
    per-(class, query) kernels put the query on gridDim.y, so the call must be cut into batches."""
    n, d, ncls = 130, 4, 3
    tr, tcls, _ = _big_training_set(n, d, ncls, 9)
    _, _, avg, _ = oracle.train_stats(tr)
    rng = np.random.default_rng(1)
    q = tr[rng.integers(0, n, 70_000)] + 1e-3 * rng.standard_normal((70_000, d))
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        pnn, _ = m.pnn_predict(q)
        knn = m.knn_predict(q, 1)
    for i in (0, 65_535, 65_536, 69_999):
        assert pnn[i] == oracle.pnn_predict(tr, tcls, avg, ncls, q[i])[0]
        assert knn[i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 1)[0]


def test_knn_vote_over_training_row_shards_inside_the_library(fir, oracle):
    """fir_cls_sharded_knn_predict: per-shard k-nearest lists per class, merged on the device and by ncclAllGather; the class is
    exactly the one-handle call's (and the oracle's), also when a class has fewer than k rows and when it straddles two shards."""
    n, d, ncls = 6000, 48, 23
    tr, tcls, q = _big_training_set(n, d, ncls, 12)
    keep = np.ones(n, bool)
    keep[np.nonzero(tcls == 5)[0][2:]] = False              # class 5 keeps two rows: it can never collect three votes
    tr, tcls = tr[keep], tcls[keep]
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        ref = {k: m.knn_predict(q, k) for k in (1, 3, 5)}
    for spd in (1, 3, 8):
        with fir.ShardedClsModel(tr, tcls, ncls, avg, devices=[0], shards_per_device=spd) as s:
            for k in (1, 3, 5):
                assert np.array_equal(s.knn_predict(q, k), ref[k]), (spd, k)
            b, _ = s.pnn_predict(q)                           # the same handle still answers the PNN
            assert b.shape == (q.shape[0],)
    for i in (0, 7, 23):
        assert ref[3][i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], 3)[0]


@pytest.mark.parametrize("step", [1, 2])
def test_a_failing_training_set_shard_fails_pnn_and_knn_instead_of_blocking(fir, fir_audit, oracle, step):
    """The sharded classifiers' failure semantics (include/fir_amd.h): shard 2 of 5 fails its step; PNN (status element behind
    the ncclSum payload) and kNN (behind the gathered tables) return its error, the handle is closed, a fresh one works."""
    n, d, ncls = 3000, 64, 13
    tr, tcls, q = _big_training_set(n, d, ncls, 6)
    _, _, avg, _ = oracle.train_stats(tr)
    for call in (lambda s: s.pnn_predict(q), lambda s: s.knn_predict(q, 3)):
        with fir_audit.ShardedClsModel(tr, tcls, ncls, avg, devices=[0], shards_per_device=5, fail_shard=3, fail_step=step, timeout_ms=20000) as s:
            with pytest.raises(fir_audit.FirError) as e:
                call(s)
            assert e.value.code == -3, (e.value.code, str(e.value))
            with pytest.raises(fir_audit.FirError) as e2:
                s.pnn_predict(q)
            assert e2.value.code == -5
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        b0, _ = m.pnn_predict(q)
        k0 = m.knn_predict(q, 3)
    with fir.ShardedClsModel(tr, tcls, ncls, avg, devices=[0], shards_per_device=5) as s:
        b, _ = s.pnn_predict(q)
        kk = s.knn_predict(q, 3)
    assert np.array_equal(kk, k0) and (b == b0).mean() > 0.9


def _mixed_training_set(n, d, ncls, seed, spread):
    """Class-major float64 rows whose classes overlap (centre + spread x noise): the nearest rows of a query belong to several classes"""
    rng = np.random.default_rng(seed)
    centres = rng.random((ncls, d))
    tcls = np.sort(rng.integers(0, ncls, n)).astype(np.int32)
    tr = centres[tcls] + spread * rng.standard_normal((n, d))
    return tr, tcls, centres, rng


@pytest.mark.parametrize("k", [1, 3, 5])
@pytest.mark.parametrize("spread", [0.004, 0.3])
def test_knn_batches_through_the_matrix_cores_give_the_exact_scans_classes(fir, oracle, k, spread):
    """VERDICT r3 item 3 (row a12): KNNClassifier::predict (classification.cpp:116-170) for a batch -- fp16 fragments of the centred
    float64 rows nominate, the reference's float64 arithmetic re-ranks, the certificate covers the rest, the vote runs over the K'
    nearest rows; what that does not settle takes the exact scan. Forced on (fir_cls_set_knn_mfma) for a 40 000 x 96 training set:
    well separated classes (every vote settles inside the nominated rows) and heavily overlapping ones (many walks need more than
    eight rows: those queries fall through), exact duplicates across classes (equal distances: never ours to order), a query that IS
    a training row, a NaN query. Classes = the exact scan's for every query, = the oracle's on a sample."""
    n, d, ncls, qb = 40_000, 96, 23, 300
    tr, tcls, centres, rng = _mixed_training_set(n, d, ncls, 31, spread)
    tr[n - 7] = tr[11]                                    # the same row in the first and in the last class: equal distances
    tr[n // 2] = tr[11]
    q = centres[rng.integers(0, ncls, qb)] + spread * rng.standard_normal((qb, d))
    q[0] = tr[11]
    q[1] = tr[777]
    q[2, 5] = np.nan
    q[3] = 0.5 * (tr[100] + tr[n - 100])                  # halfway between two classes
    _, _, avg, _ = oracle.train_stats(tr)
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        m.set_knn_mfma(0)
        exact = m.knn_predict(q, k)
        m.set_knn_mfma(1)
        got = m.knn_predict(q, k)
        st = m.knn_stats()
        again = m.knn_predict(q[:130], k)                 # a second, smaller batch on the same state (a half-filled pair)
    assert np.array_equal(got, exact)
    assert np.array_equal(again, exact[:130])
    assert st["matrix_core_queries"] == qb
    if spread < 0.01:
        assert st["exact_scan_queries_of_them"] <= 8, st  # the duplicates, the NaN query, the halfway query at most
    for i in (1, 3, 7, 150, 299):
        assert got[i] == oracle.knn_predict(tr, tcls, avg, ncls, q[i], k)[0], i


def test_knn_matrix_core_certificate_is_needed_and_a_shrunken_bound_is_caught(fir, fir_audit, oracle, monkeypatch):
    """The audit of test_gpu_gemm.py::test_the_certificate_bound_is_needed... for the float64 classifier: a coherent-rounding near-tie
    (every component of row A rounds DOWN to fp16, every component of row B UP: B's proxy is lower than A's by 0.77 of the rounding
    window although A is the query itself) between rows of two classes, avg = 0 so that the centred rows are the rows. With the bound
    as derived kNN-1 returns A's class; with FIR_GEMM_EREL_SCALE=0.25 -- honoured by the audit build only -- A falls out of the
    window, B is certified and the WRONG class comes back: the suite can see an unsound bound on this path too."""
    rng = np.random.default_rng(91)
    n, d, ncls = 6000, 512, 2
    tr = rng.random((n, d))
    a = np.full(d, 1.0 + 0.499 * 2.0 ** -10)
    b = np.full(d, 1.0 + 0.501 * 2.0 ** -10)
    ia, ib = n // 3, 2 * n // 3 + 5
    tr[ia], tr[ib] = a, b
    tcls = (np.arange(n) >= n // 2).astype(np.int32)      # class 0: the first half (A), class 1: the second (B)
    q = np.vstack([a] + [tr[i] * 0.999 for i in (7, 99, 1234)] + [rng.random(d) for _ in range(124)])
    avg = np.zeros(d)
    got = {}
    for lib_name, pkg in (("audit", fir_audit), ("shipped", fir)):
        for scale in ("1", "0.25"):
            monkeypatch.setenv("FIR_GEMM_EREL_SCALE", scale)
            with pkg.ClsModel(tr, tcls, ncls, avg, 0) as m:
                m.set_knn_mfma(1)
                got[lib_name, scale] = (m.knn_predict(q, 1), m.knn_stats())
        monkeypatch.delenv("FIR_GEMM_EREL_SCALE")
    with fir.ClsModel(tr, tcls, ncls, avg, 0) as m:
        m.set_knn_mfma(0)
        exact = m.knn_predict(q, 1)
    assert exact[0] == 0 == oracle.knn_predict(tr, tcls, avg, ncls, q[0], 1)[0]
    for key in (("audit", "1"), ("shipped", "1"), ("shipped", "0.25")):
        assert np.array_equal(got[key][0], exact), key
    wrong, st = got["audit", "0.25"]
    assert wrong[0] == 1 and st["exact_scan_queries_of_them"] <= 2, (wrong[0], st)     # B's class, certified: the shrunken bound is unsound and it shows


def test_knn_at_scale_takes_the_matrix_cores_by_default(fir, oracle):
    """300 000 x 256 float64 rows (614 MB: streams from HBM), 1 024 queries: the default dispatch nominates through the matrix cores
    (fir_cls_last_dispatch names the pass), kNN-1 / kNN-3 classes equal the exact scan's, (nearly) nothing falls through."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, d, ncls, qb = 300_000, 256, 300, 1024
    g = torch.Generator(device=dev)
    g.manual_seed(4)
    centres = torch.rand((ncls, d), generator=g, device=dev, dtype=torch.float64)
    tcls = (torch.arange(n, device=dev) * ncls // n).to(torch.int64)
    tr = centres[tcls] + 0.05 * torch.randn((n, d), generator=g, device=dev, dtype=torch.float64)
    avg = tr.mean(dim=0).cpu().numpy()
    pick = torch.randint(0, ncls, (qb,), generator=g, device=dev)
    q = (centres[pick] + 0.05 * torch.randn((qb, d), generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
    torch.cuda.synchronize()
    with fir.ClsModel(None, tcls.to(torch.int32).cpu().numpy(), ncls, avg, 0, dev_ptr=tr.data_ptr(), nt=n, d=d) as m:
        m.profile_enable(True)
        k1, k3 = m.knn_predict(q, 1), m.knn_predict(q, 3)
        disp = m.last_dispatch()
        st = m.knn_stats()
        m.set_knn_mfma(0)
        e1, e3 = m.knn_predict(q[:256], 1), m.knn_predict(q[:256], 3)
    assert "k_gemm_proxy_f16x" in disp["kernel"] and disp["flops_per_launch"] > 0, disp
    assert np.array_equal(k1[:256], e1) and np.array_equal(k3[:256], e3)
    assert np.mean(k1 == pick.cpu().numpy()) > 0.99
    assert st["matrix_core_queries"] == 2 * qb and st["exact_scan_queries_of_them"] <= qb // 16, st


def test_knn_through_the_matrix_cores_is_scale_free(fir):
    """Training rows, their average and the queries all multiplied by 2^e (exact in float64): every distance scales by 4^e, so kNN-1 and
    kNN-3 classes are the same -- through the matrix-core path the centred rows' fp16 fragments are cut with a different power-of-two
    scale and the certificate's window scales with them. e = 9 and e = -11 against e = 0, which is checked against the exact scan."""
    torch = pytest.importorskip("torch")
    dev = torch.device("cuda", 0)
    n, d, ncls, qb = 300_000, 256, 300, 512
    g = torch.Generator(device=dev)
    g.manual_seed(41)
    centres = torch.rand((ncls, d), generator=g, device=dev, dtype=torch.float64)
    tcls = (torch.arange(n, device=dev) * ncls // n).to(torch.int64)
    tr = centres[tcls] + 0.05 * torch.randn((n, d), generator=g, device=dev, dtype=torch.float64)
    pick = torch.randint(0, ncls, (qb,), generator=g, device=dev)
    q = centres[pick] + 0.05 * torch.randn((qb, d), generator=g, device=dev, dtype=torch.float64)
    cls32 = tcls.to(torch.int32).cpu().numpy()
    res = {}
    for e in (0, 9, -11):
        trs = (tr * 2.0 ** e).contiguous()
        avg = trs.mean(dim=0).cpu().numpy() if e == 0 else res["avg0"] * 2.0 ** e
        if e == 0:
            res["avg0"] = avg
        qs = (q * 2.0 ** e).cpu().numpy()
        torch.cuda.synchronize()
        with fir.ClsModel(None, cls32, ncls, avg, 0, dev_ptr=trs.data_ptr(), nt=n, d=d) as m:
            k1, k3 = m.knn_predict(qs, 1), m.knn_predict(qs, 3)
            st = m.knn_stats()
            assert st["matrix_core_queries"] == 2 * qb and st["exact_scan_queries_of_them"] <= qb // 8, st
            if e == 0:
                m.set_knn_mfma(0)
                assert np.array_equal(k1[:128], m.knn_predict(qs[:128], 1)) and np.array_equal(k3[:128], m.knn_predict(qs[:128], 3))
        res[e] = (k1, k3)
        del trs
    for e in (9, -11):
        assert np.array_equal(res[e][0], res[0][0]) and np.array_equal(res[e][1], res[0][1]), e
