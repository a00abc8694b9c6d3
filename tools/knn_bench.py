#!/usr/bin/env python3
"""kNN-1 / kNN-3 over bench.py's K3 training set (1M x 512 float64, 1000 classes) by batch size: the matrix-core path against the exact scan.
usage: python tools/knn_bench.py [--queries 4096]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
ap = argparse.ArgumentParser()
ap.add_argument("--queries", type=int, default=4096)
a = ap.parse_args()
fir = ge.load_package()
dev = torch.device("cuda", 0)
n, d, ncls, qb = 1_000_000, 512, 1000, a.queries
g = torch.Generator(device=dev); g.manual_seed(31337)
centres = torch.rand((ncls, d), generator=g, device=dev, dtype=torch.float64)
tcls = (torch.arange(n, device=dev) * ncls // n).to(torch.int64)
tr = torch.empty((n, d), device=dev, dtype=torch.float64)
for lo in range(0, n, 125_000):
    hi = lo + 125_000
    tr[lo:hi] = centres[tcls[lo:hi]] + 0.004 * torch.randn((hi - lo, d), generator=g, device=dev, dtype=torch.float64)
avg = tr.mean(dim=0).cpu().numpy()
pick = torch.randint(0, ncls, (qb,), generator=g, device=dev)
q = (centres[pick] + 0.004 * torch.randn((qb, d), generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
torch.cuda.synchronize()
m = fir.ClsModel(None, tcls.to(torch.int32).cpu().numpy(), ncls, avg, dev.index, dev_ptr=tr.data_ptr(), nt=n, d=d)
del tr
torch.cuda.empty_cache()
m.profile_enable(True)
m.set_knn_mfma(0)
ex = {k: m.knn_predict(q[:128], k) for k in (1, 3)}
m.set_knn_mfma(-1)
for k in (1, 3):
    for nq in (128, 1024, qb):
        r = m.knn_predict(q[:nq], k)
        m.profile_read()
        s0 = m.knn_stats()
        t0 = time.perf_counter()
        for _ in range(3):
            r = m.knn_predict(q[:nq], k)
        dt = (time.perf_counter() - t0) / 3
        ms, _, kname = m.profile_read()
        s1 = m.knn_stats()
        disp = m.last_dispatch()
        tf = disp["flops_per_launch"] / (float(np.min(ms)) * 1e-3) / 1e12 if len(ms) and disp["flops_per_launch"] else 0
        print(f"kNN-{k} {nq:6d} queries/call: {nq / dt:10.0f} q/s  {dt * 1e3:8.3f} ms/call  first pass launch {float(np.min(ms)) if len(ms) else 0:.3f} ms = {tf:.0f} TFLOP/s  {disp['kernel']}  "
              f"exact-scan queries/call {(s1['exact_scan_queries_of_them'] - s0['exact_scan_queries_of_them']) / 3:.1f}  = exact scan's classes on 128: {bool(np.array_equal(r[:128], ex[k]))}  "
              f"planted class found {np.mean(r == pick.cpu().numpy()[:nq]):.4f}", flush=True)
m.close()
