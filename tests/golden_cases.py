"""Input builders shared by tests/golden/make_golden.py (which asks the REAL reference for the
expected outputs) and the tests that replay the fixtures. Inputs are regenerated from seeds
(tests/synth.py); only expected outputs live in tests/golden/*.npz."""
import numpy as np

import synth

L2, CHI2, KL = 0, 1, 2
METRIC_NAMES = {L2: "l2", CHI2: "chi2", KL: "kl"}

# (seed, n, d): SURVEY.md section 8c
MATCH_SHAPES = [(13, 257, 64), (17, 1000, 256), (101, 4099, 512), (13, 600, 1280), (17, 500, 1536)]
N_QUERIES = 6


def match_case(seed, n, d, metric):
    rows = synth.make_gallery(seed, n, d, metric)
    q, _ = synth.make_queries(seed, rows, N_QUERIES, metric)
    return rows, q


DEM_IMAGE_COUNTS = (0, 40, 150)


def dem_case():
    """DirectedEnumeration (ann.cpp:270-507): rows of FEATURES_COUNT features, class labels, queries."""
    rows, q = match_case(17, 500, 1536, L2)
    return rows, synth.make_labels(500, 25), q


def special_cases():
    """name -> (rows, queries, metric): ties, 1-ulp near ties, zero rows, NaN rows, nothing-found."""
    out = {}
    rows = synth.make_gallery(21, 1500, 64, L2)
    q, pick = synth.make_queries(21, rows, 4, L2)
    for i in range(4):
        src = rows[int(pick[i])].copy()
        q[i] = src
        for dup in (17 + i, 64 * 5 + 3, 1499 - i):
            rows[dup] = src
    out["ties"] = (rows, q, L2)

    rows = synth.make_gallery(22, 1200, 128, L2)
    q, _ = synth.make_queries(22, rows, 3, L2)
    base = rows[100].copy()
    for j, r in enumerate((200, 900, 1199)):
        v = base.copy()
        v[j] = np.nextafter(v[j], np.float32(2), dtype=np.float32)
        rows[r] = v
    q[:] = base
    out["near_ties"] = (rows, q, L2)

    rows = synth.make_gallery(23, 300, 32, CHI2)
    rows[3] = 0
    rows[200] = 0
    q, _ = synth.make_queries(23, rows, 3, CHI2)
    q[0, :8] = 0
    out["zero_rows_chi2"] = (rows, q, CHI2)

    rows = synth.make_gallery(24, 300, 32, L2)
    rows[5, 3] = np.nan
    rows[77] = np.nan
    q, _ = synth.make_queries(24, rows, 3, L2)
    out["nan_rows"] = (rows, q, L2)

    rows = np.full((130, 8), 1.0e4, np.float32)
    out["nothing_found"] = (rows, np.zeros((2, 8), np.float32), L2)
    return out


def twd_case(seed=13, n=1515, d=300, n_classes=101):
    rows = synth.make_gallery(seed, n, d, L2)
    cls = synth.make_labels(n, n_classes)
    q, _ = synth.make_queries(seed, rows, 12, L2, noise=0.4)
    return rows, cls, q, n_classes


TWD_CONVENTIONAL = [(0, 0.24), (1, 0.003), (2, 0.7), (0, 0.5), (1, 1e-5), (2, 0.999)]   # ImageTesting.cpp:531-533 + both outcomes
TWD_PROPOSED = [(32, 0.7), (64, 0.7), (32, 0.95)]                                      # ImageTesting.cpp:534-535


def cls_case(seed=17, n=360, d=96, n_classes=12):
    """Double-precision dataset for classification.cpp (already L2-normalised like :829-847)."""
    x = synth.uniform01(n * d, seed).reshape(n, d).astype(np.float64)
    lab = (np.arange(n) % n_classes).astype(np.int32)
    x += 0.5 * synth.uniform01(n_classes * d, seed + 1).reshape(n_classes, d).astype(np.float64)[lab]
    x /= np.sqrt((x * x).sum(axis=1))[:, None]
    return x, lab, n_classes


def _fmt(row):
    return "".join("{:f} ".format(float(v)) for v in row)


def damaged_loader_text():
    """A feature file (FEATURES_COUNT = 1536) whose rows exercise `iss >> dfeature` (db_features.cpp:83) off the happy
    path: a line that ends early (the last value repeats), a malformed field (0 from there on), exponent notation, and
    tokens the stream does not take for numbers (nan)."""
    d = 1536
    f = synth.uniform01(4 * d, 911).reshape(4, d).astype(np.float32)
    lines = []
    lines += ["/data/a/0.jpg", "alpha", _fmt(f[0])]
    lines += ["/data/a/1.jpg", "alpha", _fmt(f[1][:100])]                                   # short line
    lines += ["/data/b/2.jpg", "beta", _fmt(f[2][:50]) + "abc " + _fmt(f[2][51:])]          # malformed field
    lines += ["/data/b/3.jpg", "beta", "1.5e-1 2E0 -3.25e+0 .5 7. +0.25 " + _fmt(f[3][6:700]) + "nan " + _fmt(f[3][701:])]
    return "\n".join(lines) + "\n"


def video_text():
    """A video-feature file in the layout loadVideos reads (video.cpp:35-96): person, #videos, per video #frames, per
    frame a name line and a feature line. Persons out of order, a name with leading blanks, one short frame line."""
    d = 1536
    f = synth.uniform01(7 * d, 913).reshape(7, d).astype(np.float32)
    f[:, 3] = 0.00005
    out, k = [], 0
    for person, videos in (("zeta", (2, 1)), ("  alpha beta", (1,)), ("mid", (1, 2))):
        out += [person, str(len(videos))]
        for nframes in videos:
            out.append(str(nframes))
            for _ in range(nframes):
                out += [f"frame_{k}.jpg", _fmt(f[k]) if k != 4 else _fmt(f[k][:900])]
                k += 1
    assert k == 7
    return "\n".join(out) + "\n"


HARNESS_FEATURES_FILE = "101_ObjectCategories_inception_resnet_v2.txt"     # FEATURES_FILE_NAME of the reference's db.h as shipped


def write_harness_features(path, signal=0.35):
    """A Caltech-like feature file for the reference's own harness (testRecognition, ImageTesting.cpp:503-548): 9 classes
    of 36-44 images (30 per class become the gallery, db_features.cpp:133), 1536 features, records interleaved."""
    d = 1536
    counts = [38, 36, 44, 40, 37, 41, 39, 36, 42]
    centers = synth.uniform01(len(counts) * d, 71).reshape(len(counts), d)
    order = [c for i in range(max(counts)) for c in range(len(counts)) if i < counts[c]]
    names, classes, feats = [], [], []
    for k, c in enumerate(order):
        names.append(f"/data/101_ObjectCategories/cat{c}/image_{k:04d}.jpg")
        classes.append(f"cat{c}")
        feats.append(signal * centers[c] + synth.uniform01(d, 5000 + k))
    synth.write_feature_file(path, names, classes, np.array(feats, np.float32))
    return counts


def harness_result_lines(stdout):
    """The lines of the harness output that carry results (names, per-test and average error / recall / unreliable
    ratio), with the wall-clock field removed."""
    import re

    keep = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith(("BF", "TWD", "Proposed TWD", "test=", "Avg error=")):
            keep.append(re.sub(r" time\(ms\)=.*$", "", line))
    return keep


CLS_HARNESS_SIGNAL = 0.2      # a harder file for testClassification1: the nearest-neighbour classifiers make errors too


def cls_harness_result_lines(stdout):
    """testClassification1's result lines (classification.cpp:1071-1082): classifier name, then fraction / db_size / error /
    sigma / recall, with the per-query time removed."""
    import re

    keep = []
    for line in stdout.splitlines():
        line = line.strip()
        if line:
            keep.append(re.sub(r" avg time\(us\)=\S+", "", line))
    return keep


def ann_harness_result_lines(stdout):
    """testANN's result lines (ann.cpp:24-81 through testSetRecognition :94-109 and the DirectedEnumeration constructor):
    sizes, the threshold line, one error / checkedPercent line per method and ratio, wall-clock fields removed."""
    import re

    keep = []
    for line in stdout.splitlines():
        line = line.strip()
        if line.startswith("init took"):
            continue
        if " error=" in line and "total_time" in line:
            line = re.sub(r" total_time \(ms\)\S+", "", line)
        keep.append(line)
    return keep


FPNN_SCALES = (1.0, 0.33)                 # classification.cpp:1002-1007
FPNN_RATIOS = (0.9, 0.99)                 # output_ratio: the default (:620) and a tighter pruning threshold


def fpnn_case2():
    """A second dataset for FPNN: 40 training rows per class -> J = 4 harmonics; 70 features -> a ragged last chunk."""
    return cls_case(seed=19, n=480, d=70, n_classes=8)


def loader_case():
    """A small feature file in the producer's format (dnn_feature_extractor.py:58-64) with
    FEATURES_COUNT = 1536 columns, classes out of order, a skipped class and sub-1e-4 values."""
    d = 1536
    classes = ["accordion", "BACKGROUND_Google", "airplanes", "accordion", "anchor", "airplanes", "257.clutter", "anchor", "accordion"]
    feats = synth.uniform01(len(classes) * d, 909).reshape(len(classes), d).astype(np.float32)
    feats[:, 5] = 0.00004      # below the 1e-4 clip
    feats[2, 7] = 0.0001       # printed as 0.000100: parses back just above/below the clip
    feats[4, :50] = 0.0
    names = [f"/data/101_ObjectCategories/{c}/image_{i:04d}.jpg" for i, c in enumerate(classes)]
    return names, classes, feats, d
