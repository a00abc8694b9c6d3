#!/usr/bin/env python3
"""Top-5 through the matrix cores, repeated 32 768-query calls against 1M x 512 (for `rocprofv3 --kernel-trace --stats`)."""
import os, sys, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
n, d, qb, k = 1_000_000, 512, 32768, int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.manual_seed(2)
x = torch.rand((n, d), device=dev); x = x / x.norm(dim=1, keepdim=True)
g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
st = torch.cuda.Stream()
q = torch.rand((qb, d), device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
keys = torch.empty(qb * max(k, 1), device=dev, dtype=torch.int64)
def f():
    if k > 1: g.search_topk_keys_dev(q.data_ptr(), qb, k, keys.data_ptr(), stream=st.cuda_stream)
    else: g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
    st.synchronize()
f(); f()
t0 = time.perf_counter()
for _ in range(4): f()
t = (time.perf_counter() - t0) / 4
print(f"top-{k}: {t*1e3:.2f} ms per {qb} queries = {qb/t:.0f} q/s; {g.last_dispatch()['kernel']}")
