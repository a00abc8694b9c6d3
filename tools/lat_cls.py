#!/usr/bin/env python3
"""Median latency of one-query calls of the float64 classifier entry points (kNN, PNN, sequential PNN) at the reference's
scale (~3 000 training rows x 256 features after PCA)."""
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
