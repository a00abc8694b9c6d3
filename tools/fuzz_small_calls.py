#!/usr/bin/env python3
"""Differential run of small calls (1..40 queries) against large galleries: the default dispatch (few-block matrix-core forms, the
one-query nomination scan) against the exact scan of the same handle, over row lengths that are resident / streamed / odd in units,
feature prefixes, planted duplicates and near rows. usage: fuzz_small_calls.py [cases=24] [seed=1]"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
kernels = {}
for c in range(cases):
    d = int(rng.choice([512, 512, 640, 768, 1024, 1280, 384, 256]))
    n = int(rng.choice([1_000_000, 1_200_003, 900_001])) if d <= 640 else int(rng.choice([600_000, 700_001]))
    qb = int(rng.integers(1, 41))
    end = d if rng.random() < 0.7 else int(rng.choice([64, 128, 256]))
    g0 = torch.Generator(device=dev); g0.manual_seed(1000 + c)
    x = torch.rand((n, d), generator=g0, device=dev)
    x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    q = torch.rand((qb, d), generator=g0, device=dev)
    src = torch.randint(0, n, (qb,), generator=g0, device=dev)
    near = rng.random(qb) < 0.5
    q[torch.from_numpy(near).to(dev)] = (x[src] * 0.97 + q * 0.03)[torch.from_numpy(near).to(dev)]
    if qb > 2:
        x[n - 7] = x[int(src[0])]                   # a duplicate of a planted row far away
        q[0] = x[int(src[0])]
    q = (q / q.norm(dim=1, keepdim=True).clamp_min(1e-30)).contiguous()
    torch.cuda.synchronize()
    st = torch.cuda.Stream()
    ka = torch.empty(qb, dtype=torch.int64, device=dev); ke = torch.empty(qb, dtype=torch.int64, device=dev)
    with torch.cuda.stream(st):
        with fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream) as g:
            for rep in range(18 if qb == 1 else 2):
                g.search_top1_keys_dev(q.data_ptr(), qb, ka.data_ptr(), 0, end, stream=st.cuda_stream)
            st.synchronize()
            disp = g.last_dispatch()
            g.set_large_batch_mfma(0)
            g.search_top1_keys_dev(q.data_ptr(), qb, ke.data_ptr(), 0, end, stream=st.cuda_stream)
            st.synchronize()
            stats = g.mfma_stats()
    same = bool(torch.equal(ka, ke))
    kernels[disp["kernel"]] = kernels.get(disp["kernel"], 0) + 1
    print(f"case {c:3d} n={n} d={d} end={end} qb={qb:2d} {disp['path']:4s} {disp['kernel']:44s} same={same} second={stats['second_pass_queries']} fallback={stats['fallback_queries']}", flush=True)
    bad += 0 if same else 1
    del x, q
    torch.cuda.empty_cache()
print("kernels seen:", kernels)
print("MISMATCHES:", bad)
sys.exit(1 if bad else 0)
