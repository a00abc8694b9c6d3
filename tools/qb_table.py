#!/usr/bin/env python3
"""Whole-call throughput of fir_search_top1_keys_dev vs query batch size (Qb in {1,8,32,256,1024}),
device-resident queries, one MI355X. Also the top-K / chi2 / KL / range-distance kernels at the same gallery.
usage: python tools/qb_table.py [--rows 1000000] [--dim 512]"""
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
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    n, d = a.rows, a.dim
    x = torch.rand((n, d), device=dev)
    x = torch.where(x < 1e-4, torch.zeros_like(x), x)          # the loader's rule (db_features.cpp:85-86)
    xl2 = x / x.norm(dim=1, keepdim=True)
    g = fir.Gallery(dev_ptr=xl2.data_ptr(), n=n, d=d, metric=0, device=0)
    x1 = x / x.sum(dim=1, keepdim=True)
    g1 = fir.Gallery(dev_ptr=x1.data_ptr(), n=n, d=d, metric=1, device=0)
    del x, xl2, x1
    torch.cuda.empty_cache()
    st = torch.cuda.Stream()
    gb = n * d * 4 / 1e9
    print(f"gallery {n} x {d} f32 ({gb:.3f} GB), MI355X; whole call incl. query transposes and key init")
    print(f"{'what':34s} {'Qb':>5s} {'ms/call':>9s} {'queries/s':>11s} {'gallery GB/s':>13s}")
    for qb in (1, 8, 16, 32, 256, 1024):
        q = torch.rand((qb, d), device=dev)
        q = (q / q.norm(dim=1, keepdim=True)).contiguous()
        keys = torch.empty(qb, device=dev, dtype=torch.int64)
        for qpp in ((8,) if qb < 16 else (8, 16)):
            g.set_tuning(qpp, 0)
            with torch.cuda.stream(st):
                t = timed(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream), 3 if qb >= 256 else 10)
            passes = -(-qb // min(qpp, max(qb, 1))) if qb >= qpp else 1
            print(f"{'L2 top-1, ' + str(qpp) + ' queries/pass':34s} {qb:5d} {t*1e3:9.3f} {qb/t:11.0f} {passes*gb/t:13.0f}")
    g.set_tuning(8, 0)
    qb = 32
    for name, gal, metric in (("chi2", g1, 1), ("KL", g1, 2)):
        gal.set_metric(metric)
        q = torch.rand((qb, d), device=dev)
        q = torch.where(q < 1e-4, torch.zeros_like(q), q)
        q = (q / q.sum(dim=1, keepdim=True)).contiguous()
        keys = torch.empty(qb * 5, device=dev, dtype=torch.int64)
        with torch.cuda.stream(st):
            t1 = timed(lambda: gal.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream), 3)
            t5 = timed(lambda: gal.search_topk_keys_dev(q.data_ptr(), qb, 5, keys.data_ptr(), stream=st.cuda_stream), 3)
        print(f"{name + ' top-1 (8 queries/pass)':34s} {qb:5d} {t1*1e3:9.3f} {qb/t1:11.0f} {(qb/8)*gb/t1:13.0f}")
        print(f"{name + ' top-5 (candidate lists)':34s} {qb:5d} {t5*1e3:9.3f} {qb/t5:11.0f} {(qb/4)*gb/t5:13.0f}")
    q = torch.rand((qb, d), device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    keys = torch.empty(qb * 5, device=dev, dtype=torch.int64)
    out = torch.empty((8, n), device=dev)
    with torch.cuda.stream(st):
        t5 = timed(lambda: g.search_topk_keys_dev(q.data_ptr(), qb, 5, keys.data_ptr(), stream=st.cuda_stream), 3)
        tr = timed(lambda: g.range_distances_dev(q.data_ptr(), 8, out.data_ptr(), 0, 64, stream=st.cuda_stream), 5)
    print(f"{'L2 top-5 (candidate lists)':34s} {qb:5d} {t5*1e3:9.3f} {qb/t5:11.0f} {(qb/4)*gb/t5:13.0f}")
    print(f"{'L2 range distances [0,64) x 8':34s} {8:5d} {tr*1e3:9.3f} {8/tr:11.0f} {gb*64/d/tr:13.0f}")


if __name__ == "__main__":
    main()
