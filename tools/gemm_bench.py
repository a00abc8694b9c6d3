#!/usr/bin/env python3
"""Throughput of the matrix-core path (fir_gemm_*) vs the exact scan on the same gallery and queries, and a
check that both return identical keys. usage: python tools/gemm_bench.py [--rows 1000000] [--dim 512] [--qb 256,1024]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--qb", default="64,256,1024")
    ap.add_argument("--precision", type=int, default=1, help="0 = f32 MFMA, 1 = bf16 split, 2 = one fp16 term")
    ap.add_argument("--identities", type=int, default=0, help="> 0: a clustered gallery, rows = identity centre * (1 + spread * noise)")
    ap.add_argument("--spread", type=float, default=0.05)
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    n, d = a.rows, a.dim
    if a.identities > 0:
        centres = torch.rand((a.identities, d), device=dev)
        x = centres[torch.arange(n, device=dev) % a.identities] * (1 + a.spread * (torch.rand((n, d), device=dev) - 0.5))
    else:
        x = torch.rand((n, d), device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
    m = fir.GemmSearch(g, a.precision)
    st = torch.cuda.Stream()
    print(f"gallery {n} x {d} f32; matrix-core path ({['f32 MFMA', 'bf16-split MFMA', 'fp16 MFMA, one term'][a.precision]} + exact re-rank + certificate) vs exact scan")
    for qb in [int(v) for v in a.qb.split(",")]:
        fresh = torch.rand((qb, d), device=dev)
        pert = x[(torch.arange(qb, device=dev) * 977 + 11) % n] + (torch.rand((qb, d), device=dev) - 0.5) * 0.05 * x.mean()
        q = torch.where((torch.arange(qb, device=dev) % 2 == 0)[:, None], fresh, pert.clamp_min(0))
        q = (q / q.norm(dim=1, keepdim=True)).contiguous()
        k1 = torch.empty(qb, device=dev, dtype=torch.int64)
        k2 = torch.empty(qb, device=dev, dtype=torch.int64)
        with torch.cuda.stream(st):
            g.set_large_batch_mfma(0)
            t_scan = timed(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, k1.data_ptr(), stream=st.cuda_stream), 3)
            s0 = m.stats()
            t_gemm = timed(lambda: m.search_top1_keys_dev(q.data_ptr(), qb, k2.data_ptr(), stream=st.cuda_stream), 3)
            s1 = m.stats()
        same = bool(torch.equal(k1, k2))
        calls = 4
        fb = (s1["fallback_queries"] - s0["fallback_queries"]) / calls
        flops = 2.0 * n * d * qb
        g.set_large_batch_mfma(-1)
        print(f"Qb={qb:5d}  scan {t_scan*1e3:8.2f} ms ({qb/t_scan:9.0f} q/s)   gemm {t_gemm*1e3:8.2f} ms ({qb/t_gemm:9.0f} q/s, "
              f"{flops/t_gemm/1e12:6.1f} TFLOP/s of the dot products)  identical keys: {same}  fallback queries/call: {fb:.1f}", flush=True)
    m.close()
    g.close()


if __name__ == "__main__":
    main()
