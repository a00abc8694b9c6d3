#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference code (oracle/_ref, built from
/root/reference by oracle/build_ref.sh) on the inputs of tests/golden_cases.py.

Run in the build container only:   python tests/golden/make_golden.py
The fixtures hold expected OUTPUTS (indices, distances, classes, decisions) plus the few
inputs that cannot be regenerated (the reference's own shuffled split). No reference source.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import golden_cases as gc  # noqa: E402
import oracle_lib  # noqa: E402
import synth  # noqa: E402


def topk_from_all(dists, k):
    """K smallest of the reference's distance vector, equal distances by row order, < 100000 only."""
    order = np.argsort(dists, kind="stable")   # NaN sort last
    idx = np.full(k, -1, np.int32)
    dd = np.full(k, 100000.0, np.float32)
    j = 0
    for o in order:
        if j >= k:
            break
        if dists[o] < np.float32(100000.0):
            idx[j] = o
            dd[j] = dists[o]
            j += 1
    return idx, dd


def match_outputs(ref, rows, q, d, starts_ends):
    db = ref.db(rows, None, 0)
    out = {}
    for (s, e) in starts_ends:
        bi, bd, t5i, t5d = [], [], [], []
        for qi in q:
            alld = db.all_distances(qi, s, e)
            if s == 0:
                b = db.recognize_image_bf(qi, e)
            else:
                b = -1
                best = np.float64(100000)
                for j, v in enumerate(alld):
                    if np.float64(v) < best:
                        best, b = np.float64(v), j
            bi.append(b)
            bd.append(alld[b] if b >= 0 else np.float32(100000.0))
            ti, td = topk_from_all(alld, 5)
            t5i.append(ti)
            t5d.append(td)
        out[f"{s}_{e}"] = dict(best_idx=np.array(bi, np.int32), best_dist=np.array(bd, np.float32),
                               top5_idx=np.array(t5i, np.int32), top5_dist=np.array(t5d, np.float32))
    db.close()
    return out


def main():
    if not oracle_lib.have_ref():
        raise SystemExit("oracle/_ref not built: run oracle/build_ref.sh (needs /root/reference)")
    refs = {m: oracle_lib.load_ref(gc.METRIC_NAMES[m]) for m in (gc.L2, gc.CHI2, gc.KL)}
    fx = {}

    # ---- match path: best index / distance / top-5 over full and prefix ranges ----
    for metric in (gc.L2, gc.CHI2, gc.KL):
        for (seed, n, d) in gc.MATCH_SHAPES:
            rows, q = gc.match_case(seed, n, d, metric)
            ranges = [(0, d)] + ([(0, 64), (64, 256)] if d >= 256 else [(0, 32), (5, 39)])
            res = match_outputs(refs[metric], rows, q, d, ranges)
            for rk, v in res.items():
                for name, arr in v.items():
                    fx[f"match/{gc.METRIC_NAMES[metric]}/{seed}_{n}_{d}/{rk}/{name}"] = arr
    for name, (rows, q, metric) in gc.special_cases().items():
        d = rows.shape[1]
        res = match_outputs(refs[metric], rows, q, d, [(0, d)])
        for nm, arr in res[f"0_{d}"].items():
            fx[f"special/{name}/{nm}"] = arr

    # ---- BruteForceClassifier (ImageTesting.cpp:58-71), BruteForce (ann.cpp:113-126), getThreshold ----
    rows, cls, q, ncls = gc.twd_case()
    db = refs[gc.L2].db(rows, cls, 0)
    for maxf in (300, 64, 256):
        got = [db.bf_classifier(qi, maxf) for qi in q]
        fx[f"bfclass/{maxf}/class"] = np.array([g[0] for g in got], np.int32)
        fx[f"bfclass/{maxf}/name"] = np.array(got[0][1])
    # ---- TWD classifiers ----
    for (typ, th) in gc.TWD_CONVENTIONAL:
        got = [db.twd_conventional(qi, ncls, typ, th, 64) for qi in q]
        fx[f"twd_conv/{typ}_{th}/class"] = np.array([g[0] for g in got], np.int32)
        fx[f"twd_conv/{typ}_{th}/unreliable"] = np.array([g[1] for g in got], np.int32)
    for (fc, th) in gc.TWD_PROPOSED:
        got = [db.twd_proposed(qi, ncls, fc, th) for qi in q]
        fx[f"twd_prop/{fc}_{th}/class"] = np.array([g[0] for g in got], np.int32)
        fx[f"twd_prop/{fc}_{th}/unreliable"] = np.array([g[1] for g in got], np.int32)
    db.close()
    rows, q = gc.match_case(17, 500, 1536, gc.L2)
    db = refs[gc.L2].db(rows, None, 1536)
    fx["ann_bf/idx"] = np.array([db.ann_bruteforce(qi) for qi in q], np.int32)
    db.close()
    # ---- DirectedEnumeration pivot table (ann.cpp:302-331): rows of FEATURES_COUNT features, class labels ----
    rows, dcls, dq = gc.dem_case()
    db = refs[gc.L2].db(rows, dcls, 1536)
    dem = db.dem(0.01, seed=13)
    piv, table, th = dem.get()
    fx["dem/pivots"], fx["dem/table"], fx["dem/threshold"] = piv, table, np.array(th, np.float32)
    for m in gc.DEM_IMAGE_COUNTS:      # DirectedEnumeration::recognize (ann.cpp:411-507) at several imageCountToCheck
        dem.set_image_count(m)
        res = [dem.recognize(q) for q in dq]
        fx[f"dem/recognize/{m}/row"] = np.array([r[0] for r in res], np.int32)
        fx[f"dem/recognize/{m}/dist"] = np.array([r[1] for r in res], np.float32)
        fx[f"dem/recognize/{m}/found"] = np.array([r[2] for r in res], np.int32)
        fx[f"dem/recognize/{m}/calc"] = np.array([r[3] for r in res], np.int32)
    dem.close()
    db.close()
    dists = synth.uniform01(1000, 55)
    for rate in (0.0, 0.01, 0.1, 0.5):
        fx[f"threshold/{rate}"] = np.array(refs[gc.L2].get_threshold(dists, rate), np.float32)

    # ---- loader + split ----
    names, classes, feats, d = gc.loader_case()
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "feats.txt")
        synth.write_feature_file(path, names, classes, feats)
        for metric in (gc.L2, gc.CHI2):
            r, c, nc = refs[metric].load_images(path)
            fx[f"loader/{gc.METRIC_NAMES[metric]}/rows"] = r
            fx[f"loader/{gc.METRIC_NAMES[metric]}/class"] = c
            fx[f"loader/{gc.METRIC_NAMES[metric]}/n_classes"] = np.array(nc, np.int32)
        # classification.cpp loader (double): reads FEATURES_FILE_NAME from the CWD
        rc = oracle_lib.load_ref("cls")
        fname = rc.L.ref_cls_features_file_name().decode()
        synth.write_feature_file(os.path.join(td, fname), names, classes, feats)
        cwd = os.getcwd()
        os.chdir(td)
        try:
            n = rc.L.ref_cls_load_dataset_cwd()
        finally:
            os.chdir(cwd)
        r = np.empty((n, rc.L.ref_cls_num_features()), np.float64)
        lab = np.empty(n, np.int32)
        rc.L.ref_cls_get_dataset(r.ctypes.data, lab.ctypes.data)
        fx["loader/f64/rows"] = r
        fx["loader/f64/labels"] = lab
        # damaged rows: what `iss >> dfeature` leaves behind on short / malformed lines
        dpath = os.path.join(td, "damaged.txt")
        with open(dpath, "w") as fh:
            fh.write(gc.damaged_loader_text())
        for metric in (gc.L2, gc.CHI2):
            r, c, nc = refs[metric].load_images(dpath)
            fx[f"loader_damaged/{gc.METRIC_NAMES[metric]}/rows"] = r
        with open(os.path.join(td, fname), "w") as fh:
            fh.write(gc.damaged_loader_text())
        os.chdir(td)
        try:
            n = rc.L.ref_cls_load_dataset_cwd()
        finally:
            os.chdir(cwd)
        r = np.empty((n, rc.L.ref_cls_num_features()), np.float64)
        lab = np.empty(n, np.int32)
        rc.L.ref_cls_get_dataset(r.ctypes.data, lab.ctypes.data)
        fx["loader_damaged/f64/rows"] = r
        # loadVideos (video.cpp:35-96): reads VIDEO_FEATURES_FILE from the CWD
        with open(os.path.join(td, refs[gc.L2].video_features_file()), "w") as fh:
            fh.write(gc.video_text())
        os.chdir(td)
        try:
            for metric in (gc.L2, gc.CHI2):
                names, vpp, fpv, rows = refs[metric].load_videos_cwd()
                pre = f"videos/{gc.METRIC_NAMES[metric]}/"
                fx[pre + "names"] = np.array(names)
                fx[pre + "videos_per_person"], fx[pre + "frames_per_video"], fx[pre + "rows"] = vpp, fpv, rows
        finally:
            os.chdir(cwd)
    counts = np.array([45, 31, 30, 29, 1, 400, 120], np.int32)
    dbi, dbc, ti, tc = refs[gc.L2].split_noshuffle(counts)
    fx["split/db_index"], fx["split/db_class"], fx["split/test_index"], fx["split/test_class"] = dbi, dbc, ti, tc

    # ---- classification.cpp kNN / PNN on an explicit (read-back) split ----
    x, lab, ncls = gc.cls_case()
    rc.set_dataset(x, lab, ncls)
    train, tcls, test = rc.split(10.0, seed=13)     # fraction >= 1 -> that many images per class (:953)
    fx["cls/train"], fx["cls/train_class"], fx["cls/test"] = train, tcls, test
    mn, mx, avg, sd = rc.stats()
    fx["cls/min"], fx["cls/max"], fx["cls/avg"], fx["cls/std"] = mn, mx, avg, sd
    for kind, param, nm in ((0, 1, "knn1"), (0, 3, "knn3"), (1, 0, "pnn"), (2, 0, "pnn_seq"), (3, 5, "pnn_clust5"), (3, 2, "pnn_clust2")):
        fx[f"cls/{nm}"] = np.array([rc.predict_row(kind, param, int(r)) for r in test], np.int32)

    # ---- FPNNClassifier (classification.cpp:618-791) on the same split (J = 3) and on a second one (J = 4) ----
    def fpnn_golden(tag, test_rows):
        for sc in gc.FPNN_SCALES:
            J, a = rc.fpnn_model(sc)
            fx[f"{tag}/{sc}/J"], fx[f"{tag}/{sc}/a"] = np.array(J, np.int32), a
            fx[f"{tag}/{sc}/bf"] = np.array([rc.fpnn_predict(sc, True, 0.9, row=int(r)) for r in test_rows], np.int32)
            for ratio in gc.FPNN_RATIOS:
                fx[f"{tag}/{sc}/seq_{ratio}"] = np.array([rc.fpnn_predict(sc, False, ratio, row=int(r)) for r in test_rows], np.int32)
    fpnn_golden("fpnn", test)
    fx["fpnn/fastlog_in"] = np.array([0.9, 0.99, 1.0, 0.5, 1e-3, 3.75, 1e-30, 7e8], np.float32)
    fx["fpnn/fastlog_out"] = np.array([rc.fastlog(v) for v in fx["fpnn/fastlog_in"]], np.float32)
    x2, lab2, ncls2 = gc.fpnn_case2()
    rc.set_dataset(x2, lab2, ncls2)
    train2, tcls2, test2 = rc.split(40.0, seed=29)
    fx["fpnn2/train"], fx["fpnn2/train_class"], fx["fpnn2/test"] = train2, tcls2, test2
    _, _, avg2, sd2 = rc.stats()
    fx["fpnn2/avg"], fx["fpnn2/std"] = avg2, sd2
    fpnn_golden("fpnn2", test2)

    out = os.path.join(HERE, "reference_outputs.npz")
    np.savez_compressed(out, **fx)
    print(f"wrote {out}: {len(fx)} arrays, {os.path.getsize(out) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
