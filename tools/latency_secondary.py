#!/usr/bin/env python3
"""Median latency of one-query calls of the secondary entry points -- the float64 classifiers (kNN, PNN, sequential PNN),
FPNN, the DEM likelihood update and the candidate-row distances of the DEM walk -- at the reference's scale (~3 000
training rows x 256 features after PCA)."""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

fir = ge.load_package()
rng = np.random.default_rng(2)
nt, d, ncls = 3030, 256, 101
x = rng.random((nt, d))
lab = np.sort(rng.integers(0, ncls, nt)).astype(np.int32)
q = rng.random((1, d))
m = fir.ClsModel(x, lab, ncls, x.mean(0), 0)
calls = {"knn_predict(1)": lambda: m.knn_predict(q, 1), "pnn_predict": lambda: m.pnn_predict(q), "pnn_predict_seq": lambda: m.pnn_predict_seq(q)}
gc.disable()
for name, fn in calls.items():
    for _ in range(100):
        fn()
    ts = []
    for _ in range(1000):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print("%-18s median %.1f us" % (name, np.median(ts) * 1e6))

# FPNN (orthogonal-series PNN) and the DEM likelihood update at the same scale
sd = x.std(0) + 1e-9
f = fir.Fpnn(x, lab, ncls, x.mean(0), sd, 1.0, 0)
rows32 = rng.random((3030, 256), dtype=np.float32)
g = fir.Gallery(rows32, (np.arange(3030) % 101).astype(np.int32), 0, 0)
dem = fir.Dem(g, 0, 45)
q32 = rng.random((1, 256), dtype=np.float32)
more = {"fpnn predict": lambda: f.predict(q), "fpnn predict_seq": lambda: f.predict_seq(q, 0.9), "dem likelihoods": lambda: dem.likelihoods(q32),
        "rows_distances(64)": lambda: g.rows_distances(q32, np.arange(64, dtype=np.int32)),
        "rows_distances(900)": lambda: g.rows_distances(q32, np.arange(900, dtype=np.int32) * 3)}
for name, fn in more.items():
    for _ in range(100):
        fn()
    ts = []
    for _ in range(1000):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    print("%-18s median %.1f us" % (name, np.median(ts) * 1e6))
