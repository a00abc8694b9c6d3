"""Live cross-check of the C restatement against the real reference build (oracle/_ref) on
fresh seeds -- beyond what the committed fixtures pin. Build container only: skipped when
oracle/_ref (or /root/reference) is absent, e.g. on the GPU box."""
import numpy as np
import pytest

import golden_cases as gc
import oracle_lib
import synth

pytestmark = pytest.mark.skipif(not oracle_lib.have_ref(), reason="oracle/_ref not built (needs /root/reference)")


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [gc.L2, gc.CHI2, gc.KL])
@pytest.mark.parametrize("seed,n,d", [(1, 333, 48), (2, 2048, 512), (3, 129, 1536), (4, 64, 200)])
def test_distances_and_argmin(oracle, metric, seed, n, d):
    ref = oracle_lib.load_ref(gc.METRIC_NAMES[metric])
    assert ref.metric == metric
    rows = synth.make_gallery(seed, n, d, metric)
    q, _ = synth.make_queries(seed, rows, 4, metric)
    db = ref.db(rows, None, 0)
    for qi in q:
        for (s, e) in ((0, d), (0, min(64, d)), (d // 3, d - 1)):
            assert np.array_equal(bits(db.all_distances(qi, s, e)), bits(oracle.all_distances(rows, qi, s, e, metric)))
        assert db.recognize_image_bf(qi, d) == oracle.recognize_bf(rows, qi, 0, d, metric)[0]
        assert bits(ref.feature_distance(qi, rows[0], 1, d - 1)) == bits(oracle.feature_distance(qi, rows[0], 1, d - 1, metric))
    db.close()


def test_default_max_features_is_features_count(oracle):
    """max_features == 0 -> FEATURES_COUNT (db_features.cpp:320-321); ann BruteForce uses the full range."""
    ref = oracle_lib.load_ref("l2")
    assert ref.features_count == 1536
    rows = synth.make_gallery(9, 200, 1536, 0)
    q, _ = synth.make_queries(9, rows, 3, 0)
    db = ref.db(rows, None, 1536)
    for qi in q:
        e = oracle.recognize_bf(rows, qi, 0, 1536, 0)[0]
        assert db.recognize_image_bf(qi, 0) == e
        assert db.ann_bruteforce(qi) == e
    db.close()


@pytest.mark.parametrize("seed", [5, 6])
def test_twd_classifiers(oracle, seed):
    ref = oracle_lib.load_ref("l2")
    rows, cls, q, ncls = gc.twd_case(seed=seed, n=808, d=280, n_classes=101)
    db = ref.db(rows, cls, 0)
    for qi in q:
        for (typ, th) in gc.TWD_CONVENTIONAL:
            assert db.twd_conventional(qi, ncls, typ, th, 64) == oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64)
        for (fc, th) in gc.TWD_PROPOSED:
            assert db.twd_proposed(qi, ncls, fc, th) == oracle.twd_proposed(rows, cls, qi, fc, th)[:2]
    db.close()


@pytest.mark.parametrize("seed,frac", [(7, 5.0), (8, 0.3)])
def test_classification_predictors(oracle, seed, frac):
    rc = oracle_lib.load_ref("cls")
    x, lab, ncls = gc.cls_case(seed=seed, n=300, d=64, n_classes=10)
    rc.set_dataset(x, lab, ncls)
    train, tcls, test = rc.split(frac, seed=seed)
    tr = x[train]
    mn, mx, avg, sd = oracle.train_stats(tr)
    rmn, rmx, ravg, rsd = rc.stats()
    assert np.array_equal(avg, ravg) and np.array_equal(sd, rsd) and np.array_equal(mn, rmn) and np.array_equal(mx, rmx)
    for r in test[:60]:
        assert rc.predict_row(0, 1, int(r)) == oracle.knn_predict(tr, tcls, avg, ncls, x[r], 1)[0]
        assert rc.predict_row(0, 3, int(r)) == oracle.knn_predict(tr, tcls, avg, ncls, x[r], 3)[0]
        assert rc.predict_row(1, 0, int(r)) == oracle.pnn_predict(tr, tcls, avg, ncls, x[r])[0]
        assert rc.predict_row(2, 0, int(r)) == oracle.pnn_predict_seq(tr, tcls, avg, ncls, x[r])[0]
