#!/usr/bin/env python3
"""One small-gallery host-pointer call repeated (for profiling a single case).
usage: python tools/lat_probe.py [rows] [dim] [queries] [variant]   variant: plain | cls | near | devfirst"""
import os
import sys
import time

import gc

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

args = sys.argv[1:] + ["3030", "1536", "8", "plain"][len(sys.argv) - 1:]
n, d, qb, variant = int(args[0]), int(args[1]), int(args[2]), args[3]
fir = ge.load_package()
if os.environ.get('NOGC'):
    gc.disable()
rng = np.random.default_rng(1)
rows = rng.random((n, d), dtype=np.float32)
rows /= np.linalg.norm(rows, axis=1, keepdims=True)
q = rng.random((qb, d), dtype=np.float32)
cls = None
if variant in ("cls", "near", "devfirst"):
    cls = (np.arange(n) % 101).astype(np.int32)
if variant in ("near", "devfirst"):
    q = rows[:qb] * np.float32(0.9) + rows[64:64 + qb] * np.float32(0.1)
g = fir.Gallery(rows, cls, 0, 0)
if variant == "devfirst":
    qd = torch.from_numpy(np.ascontiguousarray(rows[:64])).cuda()
    kd = torch.empty(64, dtype=torch.int64, device="cuda")
    st = torch.cuda.Stream()
    for _ in range(20):
        g.search_top1_keys_dev(qd.data_ptr(), 64, kd.data_ptr(), stream=st.cuda_stream)
        st.synchronize()
for _ in range(5):
    g.search_top1(q)
ts = []
for _ in range(400):
    t0 = time.perf_counter()
    g.search_top1(q)
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print(f"{n}x{d} qb={qb} {variant}: mean {ts.mean():.1f} median {np.median(ts):.1f} p99 {np.percentile(ts, 99):.1f} max {ts.max():.1f} us/call; "
      f"calls over 1 ms: {np.nonzero(ts > 1000)[0].tolist()[:10]}")
g.close()

# the same through the device-pointer entry point (no pinned staging, no publish kernel)
g = fir.Gallery(rows, cls, 0, 0)
qd = torch.from_numpy(np.ascontiguousarray(q)).cuda()
kd = torch.empty(max(qb, 1), dtype=torch.int64, device="cuda")
st = torch.cuda.Stream()
ts = []
for _ in range(600):
    t0 = time.perf_counter()
    g.search_top1_keys_dev(qd.data_ptr(), qb, kd.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print(f"  keys_dev+sync: mean {ts.mean():.1f} median {np.median(ts):.1f} max {ts.max():.1f}; calls over 1 ms: {np.nonzero(ts > 1000)[0].tolist()[:10]}")
# plain torch kernels + sync, for reference
x = torch.zeros(1024, device="cuda")
ts = []
for _ in range(2000):
    t0 = time.perf_counter()
    x.add_(1.0)
    x.mul_(1.0)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e6)
ts = np.array(ts)
print(f"  torch 2 kernels+sync: mean {ts.mean():.1f} median {np.median(ts):.1f} max {ts.max():.1f}; calls over 1 ms: {np.nonzero(ts > 1000)[0].tolist()[:10]}")
