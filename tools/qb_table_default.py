#!/usr/bin/env python3
"""whole-call throughput of fir_search_top1_keys_dev through the library's DEFAULT dispatch vs the
query batch, Qb in {1, 8, 32, 256, 1024, 4096, 32768}, device-resident queries, and the exact scan beside it (matrix cores off).
usage: python tools/qb_table_default.py [--rows 1000000] [--dim 512]"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def timed(fn, reps):
    for _ in range(3):
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
    torch.manual_seed(7)
    x = torch.rand((n, d), device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
    qmax = 32768
    q = torch.rand((qmax, d), device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    st = torch.cuda.Stream()
    gb = n * d * 4 / 1e9
    print(f"gallery {n} x {d} f32 ({gb:.3f} GB), one MI355X, device-pointer calls; bytes_alg = N*D*4 + Qb*D*4 + Qb*8 per gallery read")
    print(f"{'Qb':>6s} {'default dispatch':>44s} {'ms/call':>9s} {'queries/s':>11s} | {'exact scan q/s':>14s} {'ms/call':>9s} {'same keys':>9s}")
    for qb in (1, 8, 32, 256, 1024, 4096, 32768):
        kd = torch.empty(qb, device=dev, dtype=torch.int64)
        ks = torch.empty(qb, device=dev, dtype=torch.int64)
        reps = 20 if qb <= 256 else 5 if qb <= 4096 else 2
        with torch.cuda.stream(st):
            g.set_large_batch_mfma(-1)
            for _ in range(20 if qb < 8 else 1):          # small-saving cases build their state after a few calls
                g.search_top1_keys_dev(q.data_ptr(), qb, kd.data_ptr(), stream=st.cuda_stream)
            td = timed(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, kd.data_ptr(), stream=st.cuda_stream), reps)
            disp = g.last_dispatch()
            g.set_large_batch_mfma(0)
            ts = timed(lambda: g.search_top1_keys_dev(q.data_ptr(), qb, ks.data_ptr(), stream=st.cuda_stream), max(1, reps // 2))
        print(f"{qb:6d} {disp['kernel'][-44:]:>44s} {td * 1e3:9.3f} {qb / td:11.0f} | {qb / ts:14.0f} {ts * 1e3:9.3f} {str(bool(torch.equal(kd, ks))):>9s}", flush=True)
    g.close()


if __name__ == "__main__":
    main()
