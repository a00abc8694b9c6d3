"""Deterministic synthetic galleries and queries.

A counter-based generator (splitmix64 over the element index) written here so that every
process -- test, golden generator, bench -- produces identical bits without sharing state.
Features are Uniform[0,1) (non-negative like post-ReLU CNN features, so chi-square / KL are
defined), |x| < 1e-4 -> 0 and rows are L2-normalised (L2 metric) or L1-normalised (chi2 / KL),
mirroring qt_cpp/db_features.cpp:85-101.
"""
import numpy as np

_M64 = (1 << 64) - 1


def splitmix64(idx, seed):
    """idx: uint64 array of counters -> uint64 array of hashes."""
    with np.errstate(over="ignore"):
        z = idx.astype(np.uint64) + np.uint64((seed * 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019) & _M64)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def uniform01(n, seed, offset=0):
    """n float32 values in [0,1), element i depends only on (seed, offset + i)."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    return ((splitmix64(idx, seed) >> np.uint64(40)).astype(np.float32)) * np.float32(1.0 / (1 << 24))


def normalise(rows, metric):
    rows = np.array(rows, dtype=np.float32, copy=True)
    rows[np.abs(rows) < np.float32(1e-4)] = 0
    if metric == 0:
        s = np.sqrt((rows * rows).sum(axis=1, dtype=np.float32)).astype(np.float32)
    else:
        s = rows.sum(axis=1, dtype=np.float32)
    s[s == 0] = 1
    return (rows / s[:, None]).astype(np.float32)


def make_gallery(seed, n, d, metric=0):
    return normalise(uniform01(n * d, seed).reshape(n, d), metric)


def make_labels(n, n_classes):
    return (np.arange(n, dtype=np.int32) % n_classes).astype(np.int32)


def make_queries(seed, gallery, qb, metric=0, noise=0.05):
    """Half fresh draws, half perturbed gallery rows (so true neighbours exist)."""
    n, d = gallery.shape
    fresh = uniform01(qb * d, seed + 7919).reshape(qb, d)
    pick = (splitmix64(np.arange(qb, dtype=np.uint64), seed + 104729) % np.uint64(max(n, 1))).astype(np.int64)
    pert = gallery[pick] + np.float32(noise) * (uniform01(qb * d, seed + 1299709).reshape(qb, d) - np.float32(0.5)) * gallery[pick].mean()
    pert = np.maximum(pert, 0).astype(np.float32)
    q = np.where((np.arange(qb) % 2 == 0)[:, None], fresh, pert)
    return normalise(q, metric), pick


def write_feature_file(path, names, classes, feats):
    """The producer's text format, qt_cpp/dnn_feature_extractor.py:58-64: path, class, '{:f} '*D."""
    with open(path, "w") as f:
        for nm, cl, row in zip(names, classes, feats):
            f.write(nm + "\n")
            f.write(cl + "\n")
            f.write("".join("{:f} ".format(float(v)) for v in row))
            f.write("\n")
