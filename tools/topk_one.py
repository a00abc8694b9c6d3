#!/usr/bin/env python3
"""One form of L2 top-K at 1M x d, 32768 queries per call, 4 calls (for rocprofv3 --kernel-trace --stats). usage: topk_one.py [d] [k]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n, qb = 1_000_000, 32768
torch.manual_seed(9)
x = torch.rand((n, d), device=dev); x = x / x.norm(dim=1, keepdim=True)
q = torch.rand((qb, d), device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
torch.cuda.synchronize()
st = torch.cuda.Stream()
g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
keys = torch.empty(qb * k, device=dev, dtype=torch.int64)
for i in range(5):
    t0 = time.perf_counter()
    if k == 1: g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
    else: g.search_topk_keys_dev(q.data_ptr(), qb, k, keys.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
    print(f"call {i}: {qb / (time.perf_counter() - t0):.0f} q/s", g.last_dispatch()["kernel"], flush=True)
