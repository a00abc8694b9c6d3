#!/usr/bin/env python3
"""Median latency of one-query fir_search_top1 calls (host pointers), the reference's own call pattern.
usage: python tools/lat_one.py [rows] [dim]"""
import gc
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge  # noqa: E402

fir = ge.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3030
d = int(sys.argv[2]) if len(sys.argv) > 2 else 1536
rng = np.random.default_rng(1)
rows = rng.random((n, d), dtype=np.float32)
q = rng.random((1, d), dtype=np.float32)
g = fir.Gallery(rows, None, 0, 0)
for _ in range(200):
    g.search_top1(q)
gc.disable()
ts = []
for _ in range(2000):
    t0 = time.perf_counter()
    g.search_top1(q)
    ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e6
print("%d x %d: median %.1f us  p10 %.1f  p90 %.1f" % (n, d, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90)))
