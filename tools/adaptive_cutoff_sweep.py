#!/usr/bin/env python3
"""The sample flow (FIR_GEMM_ADAPTIVE=0: sample passes + threshold kernel + full pass) against what the library picks by itself
(FIR_GEMM_ADAPTIVE=1, the default: adaptive_for in fir_gemm.hip): us per call, both forms, one process (the knob is read when a gallery's
matrix-core state is created). usage: adaptive_cutoff_sweep.py [k=1] [d=512]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
d = int(sys.argv[2]) if len(sys.argv) > 2 else 512
torch.manual_seed(3)
print(f"k={k} d={d}:      n      qb   sample flow us (kernel)                     default us (kernel)                            ratio   second-pass / fallback queries per call (default)")
for n in (8192, 16384, 30000, 65536, 100000, 200000, 400000, 700000):
    x = torch.rand((n, d), device=dev); x = (x / x.norm(dim=1, keepdim=True)).contiguous()
    for qb in (128, 256, 1024, 4096):
        q = torch.rand((qb, d), device=dev)
        q[::2] = x[(torch.arange(qb, device=dev)[::2] * 977 + 11) % n] * 0.97 + q[::2] * 0.03
        q = (q / q.norm(dim=1, keepdim=True)).contiguous()
        keys = torch.empty(qb * k, device=dev, dtype=torch.int64)
        st = torch.cuda.Stream()
        torch.cuda.synchronize()
        res = {}
        for mode in ("0", "1"):
            os.environ["FIR_GEMM_ADAPTIVE"] = mode
            with fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream) as g:
                g.set_large_batch_mfma(1)
                call = (lambda: g.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)) if k == 1 else \
                       (lambda: g.search_topk_keys_dev(q.data_ptr(), qb, k, keys.data_ptr(), stream=st.cuda_stream))
                for _ in range(5): call()
                st.synchronize()
                s0 = g.mfma_stats()
                ts = []
                for _ in range(30):
                    t0 = time.perf_counter(); call(); st.synchronize(); ts.append(time.perf_counter() - t0)
                s1 = g.mfma_stats()
                res[mode] = (np.median(ts) * 1e6, g.last_dispatch()["kernel"], (s1["second_pass_queries"] - s0["second_pass_queries"]) / 30.0, (s1["fallback_queries"] - s0["fallback_queries"]) / 30.0, keys.clone())
        same = bool(torch.equal(res["0"][4], res["1"][4]))
        print(f"          {n:7d} {qb:6d}   {res['0'][0]:8.1f}  {res['0'][1]:36s} {res['1'][0]:8.1f}  {res['1'][1]:36s} {res['1'][0] / res['0'][0]:5.2f}x   {res['1'][2]:.1f} / {res['1'][3]:.1f}   same keys {same}", flush=True)
    del x
