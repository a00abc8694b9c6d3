#!/usr/bin/env python3
"""Chi-square top-1 / top-5 over a 1M x 512 L1-normalised gallery (BASELINE config 3): exact scan vs the two nomination forms.
usage: python tools/chi2_bench.py [--rows 1000000] [--dim 512] [--qb 256] [--metric 1|2]   (2 = KL: exact scan vs the entropy-form nomination)"""
import argparse, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge


def rate(fn, nq, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return nq * reps / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--qb", type=int, default=256)
    ap.add_argument("--metric", type=int, default=1)
    ap.add_argument("--only", default="", help="top1 | top5: time only that call, only the default nomination form (for rocprofv3 --kernel-trace --stats)")
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    torch.manual_seed(3)
    x = torch.rand((a.rows, a.dim), device=dev)
    x = torch.where(x < 1e-4, torch.zeros_like(x), x)          # the reference's loader clips |f| < 1e-4 to 0 (db_features.cpp:85-86): plain-range values
    x = x / x.sum(dim=1, keepdim=True)
    fresh = torch.rand((a.qb, a.dim), device=dev)
    fresh = torch.where(fresh < 1e-4, torch.zeros_like(fresh), fresh)
    pert = x[(torch.arange(a.qb, device=dev) * 977 + 11) % a.rows] * (1 + 0.05 * (torch.rand((a.qb, a.dim), device=dev) - 0.5))
    q = torch.where((torch.arange(a.qb, device=dev) % 2 == 0)[:, None], fresh, pert)
    q = (q / q.sum(dim=1, keepdim=True)).contiguous()
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=a.rows, d=a.dim, metric=a.metric, device=0)
    k1 = torch.empty(a.qb, device=dev, dtype=torch.int64)
    k5 = torch.empty(a.qb * 5, device=dev, dtype=torch.int64)
    ref1 = torch.empty(a.qb, device=dev, dtype=torch.int64)
    ref5 = torch.empty(a.qb * 5, device=dev, dtype=torch.int64)
    if a.only:
        fn = (lambda: g.search_top1_keys_dev(q.data_ptr(), a.qb, k1.data_ptr())) if a.only == "top1" else (lambda: g.search_topk_keys_dev(q.data_ptr(), a.qb, 5, k5.data_ptr()))
        print(f"{a.only}: {rate(fn, a.qb, 5):9.0f} q/s")
        g.close()
        return
    os.environ["FIR_NO_CHI2_NOMINATION"] = "1"
    r1 = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), a.qb, ref1.data_ptr()), a.qb, 1)
    r5 = rate(lambda: g.search_topk_keys_dev(q.data_ptr(), a.qb, 5, ref5.data_ptr()), a.qb, 1)
    print(f"exact scan        top-1 {r1:9.0f} q/s   top-5 {r5:9.0f} q/s", flush=True)
    del os.environ["FIR_NO_CHI2_NOMINATION"]
    for form, name in ((("1", "(l-r)^2 rcp(l+r) "), ("2", "harmonic form    ")) if a.metric == 1 else (("2", "entropy form     "),)):
        os.environ["FIR_CHI2_NOMINATION"] = form
        r1 = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), a.qb, k1.data_ptr()), a.qb)
        r5 = rate(lambda: g.search_topk_keys_dev(q.data_ptr(), a.qb, 5, k5.data_ptr()), a.qb)
        print(f"{name} top-1 {r1:9.0f} q/s   top-5 {r5:9.0f} q/s   identical keys: {bool(torch.equal(k1, ref1))} / {bool(torch.equal(k5, ref5))}", flush=True)
    # the row samples in the exact metric (round 4's first half) against the nomination metric (default since), interleaved
    for rnd in range(2):
        for env, name in (("1", "exact samples    "), (None, "nomination-form samples")):
            if env: os.environ["FIR_EXACT_SAMPLES"] = env
            else: os.environ.pop("FIR_EXACT_SAMPLES", None)
            r1 = rate(lambda: g.search_top1_keys_dev(q.data_ptr(), a.qb, k1.data_ptr()), a.qb)
            r5 = rate(lambda: g.search_topk_keys_dev(q.data_ptr(), a.qb, 5, k5.data_ptr()), a.qb)
            print(f"{name} top-1 {r1:9.0f} q/s   top-5 {r5:9.0f} q/s   identical keys: {bool(torch.equal(k1, ref1))} / {bool(torch.equal(k5, ref5))}", flush=True)
    g.close()


if __name__ == "__main__":
    main()
