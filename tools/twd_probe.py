#!/usr/bin/env python3
"""One TWD call repeated (for `rocprofv3 --kernel-trace --stats`). usage: python tools/twd_probe.py [rows] [dim] [queries] [conv|conv1|conv2|prop|prop_all]   (conv = posteriors, conv1 = distance difference, conv2 = distance ratio)
(prop: a query next to a gallery row, the loop ends after the first chunk; prop_all: fresh random queries, distances concentrated: no chunk leaves one class, all 8 chunks)"""
import gc
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

args = sys.argv[1:] + ["100000", "512", "1", "conv"][len(sys.argv) - 1:]
n, d, qb, which = int(args[0]), int(args[1]), int(args[2]), args[3]
gc.disable()
fir = ge.load_package()
rng = np.random.default_rng(1)
rows = rng.random((n, d), dtype=np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
cls = (np.arange(n) // 30 % 101).astype(np.int32)
q = rows[:qb] * np.float32(0.9) + rows[64:64 + qb] * np.float32(0.1)
if which == "prop_all":
    q = rng.random((qb, d), dtype=np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
g = fir.Gallery(rows, cls, 0, 0)
ctype = {"conv": (0, 0.24), "conv1": (1, 0.003), "conv2": (2, 0.7)}.get(which)
fn = (lambda: g.twd_conventional(q, 101, ctype[0], ctype[1], 64)) if ctype else (lambda: g.twd_proposed(q, 32, 0.7))
for _ in range(5):
    fn()
ts = []
for _ in range(100):
    t0 = time.perf_counter()
    fn()
    ts.append((time.perf_counter() - t0) * 1e6)
extra = ""
if ctype:
    extra = f"  unreliable {fn()[1].tolist()[:4]}  FIR_TWD_FUSED={os.environ.get('FIR_TWD_FUSED', '(auto)')}"
else:
    extra = f"  chunks used {g.twd_proposed(q, 32, 0.7)[2].tolist()[:4]}  FIR_TWD_FUSED={os.environ.get('FIR_TWD_FUSED', '(auto)')}"
print(f"{n}x{d} qb={qb} {which}: median {np.median(ts):.1f} us/call" + extra)
g.close()
