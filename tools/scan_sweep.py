#!/usr/bin/env python3
"""Kernel-variant sweep (experiments, not the bench): time the L2 top-1 scan of one library build
(FIR_AMD_LIB) over waves / queries-per-pass settings, interleaved rounds in ONE process.
usage: FIR_AMD_LIB=... python tools/scan_sweep.py [--rows 1000000] [--dim 512] [--waves 0,2048,...] [--qpp 8]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--waves", default="0")
    ap.add_argument("--qpp", default="8")
    ap.add_argument("--batch", type=int, default=0, help="queries per call (0 = queries per pass): > qpp folds passes into one launch")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--launches", type=int, default=10)
    ap.add_argument("--tag", default=os.path.basename(os.environ.get("FIR_AMD_LIB", "default")))
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    x = torch.rand((a.rows, a.dim), device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=a.rows, d=a.dim, metric=0, device=0)
    del x
    torch.cuda.empty_cache()
    configs = [(int(q), int(w)) for q in a.qpp.split(",") for w in a.waves.split(",")]
    qmax = max(max(c[0] for c in configs), a.batch)
    q = torch.rand((qmax, a.dim), device=dev)
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    keys = torch.empty(qmax, device=dev, dtype=torch.int64)
    st = torch.cuda.Stream()
    res = {c: [] for c in configs}
    for r in range(a.rounds + 1):
        for c in configs:
            g.set_tuning(c[0], c[1])
            g.profile_enable(True)
            for _ in range(a.launches):
                g.search_top1_keys_dev(q.data_ptr(), a.batch or c[0], keys.data_ptr(), stream=st.cuda_stream)
            ms, nbytes = g.profile_read()
            if r > 0:
                res[c].append((float(np.median(ms)), float(ms.min()), nbytes))
    for c in configs:
        med = np.median([v[0] for v in res[c]])
        mn = min(v[1] for v in res[c])
        nb = res[c][0][2]
        g.set_tuning(c[0], c[1])
        nqs = a.batch or c[0]
        print(f"{a.tag:22s} qpp={c[0]:2d} batch={nqs:4d} waves={c[1]:5d} median {med*1e3:8.1f} us  min {mn*1e3:8.1f} us  "
              f"{nb/med/1e6:7.1f} GB/s (median)  {nqs/med*1e3:8.0f} q/s", flush=True)
    g.close()


if __name__ == "__main__":
    main()
