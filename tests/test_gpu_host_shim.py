"""The C++ drop-in surface (fast-image-recognition_amd/host: loadImages, getTrainingAndTestImages,
recognize_image_bf, ImageInfo::distance, BruteForceClassifier, ann BruteForce) driven the way the
reference's harnesses drive it, checked against the oracle on the same feature file."""
import json
import os
import subprocess

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "fast-image-recognition_amd", "host", "host_driver")


def test_host_shim_matches_oracle_end_to_end(tmp_path, oracle):
    d = 1536                                   # FEATURES_COUNT of the reference build (db.h:86)
    counts = [37, 33, 31, 30, 36]              # 30 gallery images per class (db_features.cpp:133), the rest are queries
    names, classes, feats = [], [], []
    centers = synth.uniform01(len(counts) * d, 5).reshape(len(counts), d)
    order = [c for i in range(max(counts)) for c in range(len(counts)) if i < counts[c]]   # classes interleaved in the file
    for k, c in enumerate(order):
        names.append(f"/data/cls{c}/img{k}.jpg")
        classes.append(f"class_{c}")
        feats.append(0.6 * centers[c] + synth.uniform01(d, 1000 + k))
    feats = np.array(feats, np.float32)
    feats[3, :7] = 0.00005
    path = str(tmp_path / "features.txt")
    synth.write_feature_file(path, names, classes, feats)

    out = subprocess.run([DRIVER, path, "1536", "64", "256"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)

    rows, cls, ncls = oracle.load_images(path, d, 0)
    assert got["images"] == rows.shape[0] == sum(counts) and got["classes"] == ncls == len(counts)
    dbi, dbc, ti, tc = oracle.split(np.array(counts, np.int32), None, True)
    assert got["gallery_index"] == list(dbi) and got["gallery_class"] == list(dbc)
    assert got["query_class"] == list(tc)
    gal, q = rows[dbi], rows[ti]
    for maxf in (1536, 64, 256):
        exp = [oracle.bf_classifier(gal, dbc, qi, maxf, 0) for qi in q]
        assert got[f"bf_{maxf}_single"] == exp
        assert got[f"bf_{maxf}_batch"] == exp
        assert got[f"bf_{maxf}_name"] == f"BF, {maxf}"
    exp_rows = [oracle.recognize_bf(gal, qi, 0, d, 0)[0] for qi in q]
    assert got["ann_rows"] == exp_rows
    assert got["recognize_image_bf"] == exp_rows
    # the same gallery split into three logical shards (fir::set_devices): split + scans + RCCL exchange inside the library
    assert got["sharded_ann_rows"] == exp_rows and got["sharded_first4"] == exp_rows[:4]
    assert got["sharded_bf_256_batch"] == [oracle.bf_classifier(gal, dbc, qi, 256, 0) for qi in q]
    # a gallery row overwritten in place with query 0's features is found at once (validate-always) and by the default
    # policy once its 20 ms window has passed; restoring the row restores the answer
    assert got["edit_before"] == exp_rows[0] == got["edit_restored"]
    assert got["edit_after_always"] == got["edit_row"] == got["edit_after_default"]
    assert np.float32(got["dist_q0_g0"]) == oracle.feature_distance(q[0], gal[0], 0, d, 0)
    assert np.float32(got["dist_q0_g0_64"]) == oracle.feature_distance(q[0], gal[0], 0, 64, 0)
    acc = np.mean(np.array(got["bf_1536_batch"]) == np.array(got["query_class"]))
    assert acc > 0.9
    # the TWD classifiers of testRecognition (ImageTesting.cpp:530-535)
    ncls = len(counts)
    for key, fn in (("twd_post", lambda qi: oracle.twd_conventional(gal, dbc, qi, ncls, 0, 0.24, 64)),
                    ("twd_diff", lambda qi: oracle.twd_conventional(gal, dbc, qi, ncls, 1, 0.003, 64)),
                    ("twd_ratio", lambda qi: oracle.twd_conventional(gal, dbc, qi, ncls, 2, 0.7, 64)),
                    ("twd_p32", lambda qi: oracle.twd_proposed(gal, dbc, qi, 32, 0.7)[:2]),
                    ("twd_p64", lambda qi: oracle.twd_proposed(gal, dbc, qi, 64, 0.7)[:2])):
        exp = [fn(qi) for qi in q]
        assert got[key + "_batch"] == [e[0] for e in exp], key
        assert got[key + "_first6"] == [e[0] for e in exp][:6], key
        assert got[key + "_unreliable"] == sum(e[1] for e in exp), key
    assert got["twd_post_name"] == "TWD posteriors, 0.24" and got["twd_p32_name"] == "Proposed TWD, 32, 0.7"
    # DirectedEnumeration (ann.cpp:270-507): the greedy pivots, the false-accept threshold and the walk
    npiv = max(5, int(len(gal) * 0.015))
    piv, table, mo = oracle.dem_pivot_table(gal, dbc, got["dem_pivots"][0], npiv, 0)
    assert got["dem_pivots"] == list(piv)
    th = oracle.get_threshold(mo, 0.01)
    assert np.float32(got["dem_threshold"]) == th
    for m in (0, 40, 100):
        exp = [oracle.dem_recognize(gal, piv, table, th, m, qi, 0) for qi in q]
        assert got[f"dem_{m}_batch"] == [e[0] for e in exp], m
        assert got[f"dem_{m}_single"] == [e[0] for e in exp], m
        assert got[f"dem_{m}_calc"] == [e[3] for e in exp], m
        assert got[f"dem_{m}_found"] == [e[2] for e in exp], m


def test_classification_shim_matches_oracle_end_to_end(tmp_path, oracle):
    """classification.cpp-side API (load_image_dataset, split_train_test, KNNClassifier, PNNClassifier)."""
    import golden_cases as gc

    cls_driver = os.path.join(ROOT, "fast-image-recognition_amd", "host", "cls_driver")
    d, ncls = 96, 7
    x, lab, _ = gc.cls_case(seed=23, n=210, d=d, n_classes=ncls)
    names = [f"/data/c{int(c)}/i{i}.jpg" for i, c in enumerate(lab)]
    classes = [f"class_{int(c)}" for c in lab]
    path = str(tmp_path / "features.txt")
    synth.write_feature_file(path, names, classes, x.astype(np.float32))
    out = subprocess.run([cls_driver, path, str(d), "12"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    got = json.loads(out.stdout)

    rows, labels, nc = oracle.load_dataset_f64(path, d)
    assert (got["classes"], got["features"], got["rows"]) == (nc, d, rows.shape[0]) == (ncls, d, 210)
    train = np.concatenate([np.nonzero(labels == c)[0][:12] for c in range(nc)])      # fraction >= 1: that many per class (:953)
    test = np.concatenate([np.nonzero(labels == c)[0][12:] for c in range(nc)])
    assert got["train_rows"] == list(train) and got["test_rows"] == list(test)
    tr, tcls = rows[train], labels[train]
    _, _, avg, sd = oracle.train_stats(tr)
    assert got["avg0"] == avg[0]
    e1 = [oracle.knn_predict(tr, tcls, avg, nc, rows[r], 1)[0] for r in test]
    e3 = [oracle.knn_predict(tr, tcls, avg, nc, rows[r], 3)[0] for r in test]
    ep = [oracle.pnn_predict(tr, tcls, avg, nc, rows[r])[0] for r in test]
    assert got["knn1_single"] == e1 and got["knn1_batch"] == e1 and got["knn1_name"] == "k-NN, 1"
    assert got["knn3_single"] == e3 and got["knn3_batch"] == e3
    assert got["pnn_single"] == ep and got["pnn_batch"] == ep and got["pnn_name"] == "PNN"
    assert np.mean(np.array(e1) == np.array(got["truth"])) > 0.5
    es = [oracle.pnn_predict_seq(tr, tcls, avg, nc, rows[r])[0] for r in test]
    assert got["pnn_seq_single"] == es and got["pnn_seq_batch"] == es and got["pnn_seq_name"] == "PNN (seq)"
    keep = oracle.pnn_cluster_train(tr, tcls, nc, 5)            # positions in the class-major training list
    assert got["medoid_rows"] == list(train[keep])
    ec = [oracle.pnn_predict_den(tr[keep], tcls[keep], avg, nc, rows[r], tr.shape[0])[0] for r in test]
    assert got["pnn_clust5_single"] == ec and got["pnn_clust5_batch"] == ec and got["pnn_clust5_name"] == "PNN with clustering, 5"
    # FPNNClassifier (classification.cpp:618-791), the four instances of testClassification1 (:1002-1007)
    for key, sc, seq, ratio, name in (("fpnn", 1.0, False, 0.9, "FPNN, 1"), ("fpnn033", 0.33, False, 0.9, "FPNN, 0.33"),
                                      ("fpnn_seq", 1.0, True, 0.9, "FPNN, 1 (seq)"), ("fpnn033_seq", 0.33, True, 0.99, "FPNN, 0.33 (seq)")):
        J, a = oracle.fpnn_train(tr, tcls, nc, avg, sd, sc)
        assert got["fpnn_J"] == J
        ef = [oracle.fpnn_predict(a, J, nc, avg, sd, sc, rows[r], seq, ratio)[0] for r in test]
        assert got[key + "_single"] == ef and got[key + "_batch"] == ef, key
        assert got[key + "_name"] == name
