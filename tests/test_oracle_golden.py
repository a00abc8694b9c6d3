"""Pins the oracle (oracle/oracle.c) to the reference: every fixture in
tests/golden/reference_outputs.npz was produced by the REAL reference code
(tests/golden/make_golden.py over oracle/_ref); the C restatement must reproduce it exactly.
Runs anywhere (no GPU, no /root/reference)."""
import os
import tempfile

import numpy as np
import pytest

import golden_cases as gc
import synth

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def topk_all(oracle, rows, q, s, e, k, metric):
    idx, dd = [], []
    for qi in q:
        i, d_ = oracle.topk(rows, qi, s, e, k, metric)
        idx.append(i)
        dd.append(d_)
    return np.array(idx, np.int32), np.array(dd, np.float32)


@pytest.mark.parametrize("metric", [gc.L2, gc.CHI2, gc.KL])
@pytest.mark.parametrize("shape", gc.MATCH_SHAPES)
def test_match_path_matches_reference(oracle, metric, shape):
    seed, n, d = shape
    rows, q = gc.match_case(seed, n, d, metric)
    ranges = [(0, d)] + ([(0, 64), (64, 256)] if d >= 256 else [(0, 32), (5, 39)])
    for (s, e) in ranges:
        pre = f"match/{gc.METRIC_NAMES[metric]}/{seed}_{n}_{d}/{s}_{e}/"
        idx, dist = oracle.top1_batch(rows, q, s, e, metric)
        assert np.array_equal(idx, GOLD[pre + "best_idx"])
        assert np.array_equal(bits(dist), bits(GOLD[pre + "best_dist"]))   # KL too: same libm on both sides
        ti, td = topk_all(oracle, rows, q, s, e, 5, metric)
        assert np.array_equal(ti, GOLD[pre + "top5_idx"])
        assert np.array_equal(bits(td), bits(GOLD[pre + "top5_dist"]))


@pytest.mark.parametrize("name", sorted(gc.special_cases().keys()))
def test_special_cases_match_reference(oracle, name):
    rows, q, metric = gc.special_cases()[name]
    d = rows.shape[1]
    idx, dist = oracle.top1_batch(rows, q, 0, d, metric)
    assert np.array_equal(idx, GOLD[f"special/{name}/best_idx"])
    assert np.array_equal(bits(dist), bits(GOLD[f"special/{name}/best_dist"]))
    ti, td = topk_all(oracle, rows, q, 0, d, 5, metric)
    assert np.array_equal(ti, GOLD[f"special/{name}/top5_idx"])
    assert np.array_equal(bits(td), bits(GOLD[f"special/{name}/top5_dist"]))


def test_bruteforce_classifier_and_twd_match_reference(oracle):
    rows, cls, q, ncls = gc.twd_case()
    for maxf in (300, 64, 256):
        got = [oracle.bf_classifier(rows, cls, qi, maxf) for qi in q]
        assert got == list(GOLD[f"bfclass/{maxf}/class"])
    assert str(GOLD["bfclass/64/name"]) == "BF, 64"
    both = set()
    for (typ, th) in gc.TWD_CONVENTIONAL:
        got = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64) for qi in q]
        assert [g[0] for g in got] == list(GOLD[f"twd_conv/{typ}_{th}/class"])
        assert [g[1] for g in got] == list(GOLD[f"twd_conv/{typ}_{th}/unreliable"])
        both.update(g[1] for g in got)
    assert both == {0, 1}, "fixtures must exercise both the reliable and the second-stage outcome"
    for (fc, th) in gc.TWD_PROPOSED:
        got = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in q]
        assert [g[0] for g in got] == list(GOLD[f"twd_prop/{fc}_{th}/class"])
        assert [g[1] for g in got] == list(GOLD[f"twd_prop/{fc}_{th}/unreliable"])


def test_ann_bruteforce_and_threshold_match_reference(oracle):
    rows, q = gc.match_case(17, 500, 1536, gc.L2)
    idx, _ = oracle.top1_batch(rows, q, 0, 1536, gc.L2)
    assert np.array_equal(idx, GOLD["ann_bf/idx"])
    dists = synth.uniform01(1000, 55)
    for rate in (0.0, 0.01, 0.1, 0.5):
        assert bits(oracle.get_threshold(dists, rate)) == bits(GOLD[f"threshold/{rate}"])


def test_fpnn_matches_reference(oracle):
    """FPNNClassifier (classification.cpp:618-791): fasterlog2, the trained coefficients and both predict forms."""
    for vin, vout in zip(GOLD["fpnn/fastlog_in"], GOLD["fpnn/fastlog_out"]):
        assert bits(oracle.fastlog(vin)) == bits(vout)
    x, lab, ncls = gc.cls_case()
    x2, lab2, ncls2 = gc.fpnn_case2()
    cases = (("fpnn", x, ncls, GOLD["cls/train"], GOLD["cls/train_class"], GOLD["cls/test"], GOLD["cls/avg"], GOLD["cls/std"], 3),
             ("fpnn2", x2, ncls2, GOLD["fpnn2/train"], GOLD["fpnn2/train_class"], GOLD["fpnn2/test"], GOLD["fpnn2/avg"], GOLD["fpnn2/std"], 4))
    for tag, xx, nc, train, tcls, test, avg, sd, wantJ in cases:
        pruned = 0
        for sc in gc.FPNN_SCALES:
            J, a = oracle.fpnn_train(xx[train], tcls, nc, avg, sd, sc)
            assert J == int(GOLD[f"{tag}/{sc}/J"]) == wantJ
            assert np.array_equal(a.view(np.uint64), GOLD[f"{tag}/{sc}/a"].view(np.uint64)), (tag, sc)
            bf = [oracle.fpnn_predict(a, J, nc, avg, sd, sc, xx[r])[0] for r in test]
            assert bf == list(GOLD[f"{tag}/{sc}/bf"]), (tag, sc)
            for ratio in gc.FPNN_RATIOS:
                res = [oracle.fpnn_predict(a, J, nc, avg, sd, sc, xx[r], True, ratio) for r in test]
                assert [r[0] for r in res] == list(GOLD[f"{tag}/{sc}/seq_{ratio}"]), (tag, sc, ratio)
                pruned += sum(r[2] < -(-xx.shape[1] // 32) for r in res)
        assert pruned > 0, "the sequential form should stop early for some queries"


def test_dem_pivot_table_matches_reference(oracle):
    """DirectedEnumeration's constructor (ann.cpp:270-348): table, greedy pivots, false-accept threshold."""
    rows, cls, _ = gc.dem_case()
    gp, gt, gth = GOLD["dem/pivots"], GOLD["dem/table"], GOLD["dem/threshold"]
    assert gp.size == max(5, int(500 * 0.015)) == 7
    piv, table, mo = oracle.dem_pivot_table(rows, cls, int(gp[0]), gp.size, gc.L2)
    assert np.array_equal(piv, gp)
    assert np.array_equal(bits(table), bits(gt))
    assert bits(oracle.get_threshold(mo, 0.01)) == bits(gth)      # otherClassesDists -> getThreshold (ann.cpp:341-343)


def test_dem_recognize_matches_reference(oracle):
    """DirectedEnumeration::recognize (ann.cpp:411-507): row, bestDistance, isFoundLessThreshold, distanceCalcCount."""
    rows, cls, queries = gc.dem_case()
    gp, gt, gth = GOLD["dem/pivots"], GOLD["dem/table"], GOLD["dem/threshold"]
    seen_found = seen_full = 0
    for m in gc.DEM_IMAGE_COUNTS:
        for i, q in enumerate(queries):
            r, bd, fo, cc = oracle.dem_recognize(rows, gp, gt, gth, m, q, gc.L2)
            want = tuple(GOLD[f"dem/recognize/{m}/{k}"][i] for k in ("row", "dist", "found", "calc"))
            assert (r, bits(bd), fo, cc) == (want[0], bits(want[1]), want[2], want[3]), (m, i)
            seen_found += fo
            seen_full += (cc == (m or len(rows)))
    assert seen_found and seen_full       # both exits of the walk are exercised


def test_loader_and_split_match_reference(oracle):
    names, classes, feats, d = gc.loader_case()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "feats.txt")
        synth.write_feature_file(path, names, classes, feats)
        for metric in (gc.L2, gc.CHI2):
            rows, cls, ncls = oracle.load_images(path, d, metric)
            pre = f"loader/{gc.METRIC_NAMES[metric]}/"
            assert ncls == int(GOLD[pre + "n_classes"]) == 3
            assert np.array_equal(cls, GOLD[pre + "class"])
            assert np.array_equal(bits(rows), bits(GOLD[pre + "rows"]))
        r64, lab, ncls = oracle.load_dataset_f64(path, d)
        assert np.array_equal(lab, GOLD["loader/f64/labels"])
        assert np.array_equal(r64.view(np.uint64), GOLD["loader/f64/rows"].view(np.uint64))
        assert oracle.load_images(os.path.join(td, "missing.txt"), d, gc.L2)[0].shape[0] == 0   # db_features.cpp:49,115
        # short / malformed feature lines: the stream's leftovers (db_features.cpp:83, classification.cpp:832)
        dpath = os.path.join(td, "damaged.txt")
        with open(dpath, "w") as fh:
            fh.write(gc.damaged_loader_text())
        for metric in (gc.L2, gc.CHI2):
            rows, _, _ = oracle.load_images(dpath, d, metric)
            assert np.array_equal(bits(rows), bits(GOLD[f"loader_damaged/{gc.METRIC_NAMES[metric]}/rows"])), metric
        r64, _, _ = oracle.load_dataset_f64(dpath, d)
        assert np.array_equal(r64.view(np.uint64), GOLD["loader_damaged/f64/rows"].view(np.uint64))
        # loadVideos (video.cpp:35-96)
        vpath = os.path.join(td, "videos.txt")
        with open(vpath, "w") as fh:
            fh.write(gc.video_text())
        for metric in (gc.L2, gc.CHI2):
            names, vpp, fpv, rows = oracle.load_videos(vpath, d, metric)
            pre = f"videos/{gc.METRIC_NAMES[metric]}/"
            assert names == list(GOLD[pre + "names"]) == ["alpha beta", "mid", "zeta"]
            assert np.array_equal(vpp, GOLD[pre + "videos_per_person"]) and np.array_equal(fpv, GOLD[pre + "frames_per_video"])
            assert np.array_equal(bits(rows), bits(GOLD[pre + "rows"])), metric
    counts = np.array([45, 31, 30, 29, 1, 400, 120], np.int32)
    dbi, dbc, ti, tc = oracle.split(counts, None, True)
    assert np.array_equal(dbi, GOLD["split/db_index"]) and np.array_equal(dbc, GOLD["split/db_class"])
    assert np.array_equal(ti, GOLD["split/test_index"]) and np.array_equal(tc, GOLD["split/test_class"])


def test_classification_knn_pnn_match_reference(oracle):
    x, lab, ncls = gc.cls_case()
    train, tcls, test = GOLD["cls/train"], GOLD["cls/train_class"], GOLD["cls/test"]
    tr = x[train]
    mn, mx, avg, sd = oracle.train_stats(tr)
    for got, nm in ((mn, "min"), (mx, "max"), (avg, "avg"), (sd, "std")):
        assert np.array_equal(got.view(np.uint64), GOLD[f"cls/{nm}"].view(np.uint64)), nm
    knn1 = [oracle.knn_predict(tr, tcls, avg, ncls, x[r], 1)[0] for r in test]
    knn3 = [oracle.knn_predict(tr, tcls, avg, ncls, x[r], 3)[0] for r in test]
    pnn = [oracle.pnn_predict(tr, tcls, avg, ncls, x[r])[0] for r in test]
    seq = [oracle.pnn_predict_seq(tr, tcls, avg, ncls, x[r])[0] for r in test]
    assert knn1 == list(GOLD["cls/knn1"])
    assert knn3 == list(GOLD["cls/knn3"])
    assert pnn == list(GOLD["cls/pnn"])
    assert seq == list(GOLD["cls/pnn_seq"])
    for k in (5, 2):        # PNNwithClusteringClassifier (classification.cpp:311-428): k-medoids per class, PNN over the medoids
        keep = oracle.pnn_cluster_train(tr, tcls, ncls, k)
        assert keep.size == ncls * k
        got = [oracle.pnn_predict_den(tr[keep], tcls[keep], avg, ncls, x[r], tr.shape[0])[0] for r in test]
        assert got == list(GOLD[f"cls/pnn_clust{k}"])
    acc = np.mean(np.array(knn1) == lab[test])
    assert acc > 0.5, "the synthetic classes should be separable enough for the fixture to mean something"
