#!/usr/bin/env python3
"""Repeated batched top-1 calls against a cache-resident gallery (BASELINE configs[1]: 100 000 x 512), for
`rocprofv3 --kernel-trace`: the timeline of one call shows where its fixed cost sits.
usage: python tools/small_call_trace.py [rows] [dim] [queries] [calls]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

args = sys.argv[1:] + ["100000", "512", "4096", "30"][len(sys.argv) - 1:]
n, d, qb, calls = (int(a) for a in args[:4])
fir = ge.load_package()
dev = torch.device("cuda", 0)
torch.manual_seed(5)
x = torch.rand((n, d), device=dev)
x = x / x.norm(dim=1, keepdim=True)
q = torch.rand((qb, d), device=dev)
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
keys = torch.empty(qb, device=dev, dtype=torch.int64)
st = torch.cuda.Stream()
ts = []
for i in range(calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(st):
        g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
print(f"{n}x{d}, {qb} queries per call: median {np.median(ts[5:]) * 1e6:.1f} us per call = {qb / np.median(ts[5:]):.0f} q/s; dispatch {g.last_dispatch()['kernel']}")
g.close()
