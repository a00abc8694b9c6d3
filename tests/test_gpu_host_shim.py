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
    assert np.float32(got["dist_q0_g0"]) == oracle.feature_distance(q[0], gal[0], 0, d, 0)
    assert np.float32(got["dist_q0_g0_64"]) == oracle.feature_distance(q[0], gal[0], 0, 64, 0)
    acc = np.mean(np.array(got["bf_1536_batch"]) == np.array(got["query_class"]))
    assert acc > 0.9
