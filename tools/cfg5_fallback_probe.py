#!/usr/bin/env python3
"""Small batches of bench.py's config-5 data (1M x 1280) through the default dispatch, many calls each: fallback queries per
call, which kernel ran, call time. FIR_GEMM_DEBUG_COUNTS=1 adds the appended rows / bound of every call on stderr.
usage: cfg5_fallback_probe.py [d=1280] [calls=20] [n=1000000]"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
fir = ge.load_package()
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 1280
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
qmax = 4096
rows = torch.empty((n, d), device=dev)
for c in range((n + bench.CHUNK_ROWS - 1) // bench.CHUNK_ROWS):
    r0 = c * bench.CHUNK_ROWS
    rows[r0:r0 + bench.CHUNK_ROWS] = bench.gen_chunk(c + 7000, min(bench.CHUNK_ROWS, n - r0), d, dev)
gq = torch.Generator(device=dev); gq.manual_seed(5151)
fresh = torch.rand((qmax, d), generator=gq, device=dev)
planted = (torch.arange(qmax, device=dev) * 977 + 11) % n
pert = (rows[planted] + (torch.rand((qmax, d), generator=gq, device=dev) - 0.5) * 0.05 * rows[:4096].mean()).clamp_min(0)
q = torch.where((torch.arange(qmax, device=dev) % 2 == 0)[:, None], fresh, pert)
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    g = fir.Gallery(dev_ptr=rows.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
    for qb in (8, 32, 64, 256, 1024):
        ks = torch.empty(qb, device=dev, dtype=torch.int64)
        kd = torch.empty(qb, device=dev, dtype=torch.int64)
        g.set_large_batch_mfma(0)
        g.search_top1_keys_dev(q.data_ptr(), qb, ks.data_ptr(), stream=st.cuda_stream)
        torch.cuda.synchronize()
        g.set_large_batch_mfma(-1)
        fbs, ts, bad = [], [], 0
        sp0 = g.mfma_stats()["second_pass_queries"]
        for i in range(calls):
            f0 = g.mfma_stats()["fallback_queries"]
            t0 = time.perf_counter()
            g.search_top1_keys_dev(q.data_ptr(), qb, kd.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e6)
            fbs.append(g.mfma_stats()["fallback_queries"] - f0)
            bad += 0 if torch.equal(ks, kd) else 1
        dd = g.last_dispatch()
        ts.sort()
        print(f"qb {qb:5d}  kernel {dd['kernel']}  fallbacks per call {fbs}  second-pass queries {g.mfma_stats()['second_pass_queries'] - sp0}  wrong-key calls {bad}  us/call median {ts[len(ts)//2]:.0f} max {ts[-1]:.0f}", flush=True)
    g.close()
