#!/usr/bin/env python3
"""Appended rows per query of the adaptive flow for a small batch (one pair, 256 row ranges) -- FIR_GEMM_DEBUG_COUNTS=1 prints them."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
n, d = 1_000_000, int(sys.argv[1]) if len(sys.argv) > 1 else 1280
qb = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.manual_seed(5)
x = torch.rand((n, d), device=dev); x = x / x.norm(dim=1, keepdim=True)
q = torch.rand((qb, d), device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
m = fir.GemmSearch(g, 2)
k = torch.empty(qb, device=dev, dtype=torch.int64)
for i in range(2):
    m.search_top1_keys_dev(q.data_ptr(), qb, k.data_ptr())
    torch.cuda.synchronize()
print("fallbacks", m.stats()["fallback_queries"])
