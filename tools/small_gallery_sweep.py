#!/usr/bin/env python3
"""Queries-per-pass sweep for cache-resident galleries (few tiles): device-resident queries, whole call + sync.
usage: python tools/small_gallery_sweep.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream()
    print(f"{'gallery':>14s} {'Qb':>5s} {'qpp':>4s} {'us/call':>9s} {'queries/s':>11s}")
    for n, d in ((3030, 1536), (3030, 512), (20000, 512), (100_000, 512)):
        x = torch.rand((n, d), device=dev)
        x = (x / x.norm(dim=1, keepdim=True)).contiguous()
        g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
        for qb in (1, 8, 64, 512):
            q = torch.rand((qb, d), device=dev)
            q = (q / q.norm(dim=1, keepdim=True)).contiguous()
            keys = torch.empty(qb, device=dev, dtype=torch.int64)
            ref = None
            for qpp in (-1, 1, 2, 4, 8, 16):
                if qpp > qb and qpp > 0:
                    continue
                g.set_tuning(qpp, 0)

                def call():
                    g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
                    st.synchronize()
                for _ in range(5):
                    call()
                reps = 50
                t0 = time.perf_counter()
                for _ in range(reps):
                    call()
                t = (time.perf_counter() - t0) / reps
                k = keys.cpu().numpy().copy()
                if ref is None:
                    ref = k
                assert np.array_equal(ref, k), "answers must not depend on the tuning"
                print(f"{str(n) + 'x' + str(d):>14s} {qb:5d} {qpp:4d} {t * 1e6:9.1f} {qb / t:11.0f}")
        g.close()
        del x


if __name__ == "__main__":
    main()
