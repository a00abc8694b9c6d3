#!/usr/bin/env python3
"""bench.py's config-5 data (1M x 1280) through one default-dispatch call of 32768 queries; prints the fallback count."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
fir = ge.load_package()
dev = torch.device("cuda", 0)
n, d, qmax = 1_000_000, 1280, 32768
rows = torch.empty((n, d), device=dev)
for c in range(64):
    rows[c * bench.CHUNK_ROWS:(c + 1) * bench.CHUNK_ROWS] = bench.gen_chunk(c + 7000, bench.CHUNK_ROWS, d, dev)
gq = torch.Generator(device=dev); gq.manual_seed(5151)
fresh = torch.rand((qmax, d), generator=gq, device=dev)
planted = (torch.arange(qmax, device=dev) * 977 + 11) % n
pert = (rows[planted] + (torch.rand((qmax, d), generator=gq, device=dev) - 0.5) * 0.05 * rows[:4096].mean()).clamp_min(0)
q = torch.where((torch.arange(qmax, device=dev) % 2 == 0)[:, None], fresh, pert)
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
g = fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=0, device=0)
k = torch.empty(qmax, device=dev, dtype=torch.int64)
for i in range(2):
    f0 = g.mfma_stats()["fallback_queries"]
    g.search_top1_keys_dev(q.data_ptr(), qmax, k.data_ptr())
    torch.cuda.synchronize()
    print("call", i, "fallback queries", g.mfma_stats()["fallback_queries"] - f0, flush=True)
print("--- bench sequence ---", flush=True)
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for qb in (8, 32, 256, 4096, 32768):
        ks = torch.empty(qb, device=dev, dtype=torch.int64)
        km = torch.empty(qb, device=dev, dtype=torch.int64)
        kd = torch.empty(qb, device=dev, dtype=torch.int64)
        g.set_large_batch_mfma(0)
        g.search_top1_keys_dev(q.data_ptr(), qb, ks.data_ptr(), stream=st.cuda_stream)
        g.set_large_batch_mfma(1)
        for i in range(3):
            f0 = g.mfma_stats()["fallback_queries"]
            g.search_top1_keys_dev(q.data_ptr(), qb, km.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            print("qb", qb, "forced call", i, "fallbacks", g.mfma_stats()["fallback_queries"] - f0, "same", bool(torch.equal(ks, km)), flush=True)
        g.set_large_batch_mfma(-1)
        for i in range(3):
            f0 = g.mfma_stats()["fallback_queries"]
            g.search_top1_keys_dev(q.data_ptr(), qb, kd.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            print("qb", qb, "default call", i, "fallbacks", g.mfma_stats()["fallback_queries"] - f0, "same", bool(torch.equal(ks, kd)), flush=True)
