#!/usr/bin/env python3
"""L2 top-5 through the matrix cores: the in-flight K-slot threshold (k_gemm_proxy_f16x<4, *>) against the sample flow
(FIR_GEMM_ADAPTIVE_TOPK=0), separate galleries in one process, interleaved rounds; identical keys to the exact top-K scan on a prefix.
usage: python tools/topk_ab.py [dim=512] [k=5]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
d = int(sys.argv[1]) if len(sys.argv) > 1 else 512
k = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n = 1_000_000
torch.manual_seed(9)
x = torch.rand((n, d), device=dev); x = x / x.norm(dim=1, keepdim=True)
qmax = 32768
q = torch.rand((qmax, d), device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
q[1::2] = x[(torch.arange(qmax // 2, device=dev) * 977 + 11) % n] * (1 + 0.02 * (torch.rand((qmax // 2, d), device=dev) - 0.5))
q = (q / q.norm(dim=1, keepdim=True)).contiguous()
torch.cuda.synchronize()
st = torch.cuda.Stream()
gs = {}
for name, env in (("slots", "1"), ("sample", "0")):
    os.environ["FIR_GEMM_ADAPTIVE_TOPK"] = env
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
    kk = torch.empty(256 * k, device=dev, dtype=torch.int64)
    g.search_topk_keys_dev(q.data_ptr(), 256, k, kk.data_ptr(), stream=st.cuda_stream); st.synchronize()     # builds the state under this knob
    gs[name] = g
ge_ = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0, stream=st.cuda_stream)
ge_.set_large_batch_mfma(0)
for qb in (256, 4096, 32768):
    res = {}
    keys = {nm: torch.empty(qb * k, device=dev, dtype=torch.int64) for nm in gs}
    for rnd in range(3):
        for nm, g in gs.items():
            s0 = g.mfma_stats()
            t0 = time.perf_counter()
            for _ in range(3):
                g.search_topk_keys_dev(q.data_ptr(), qb, k, keys[nm].data_ptr(), stream=st.cuda_stream)
            st.synchronize()
            t = (time.perf_counter() - t0) / 3
            s1 = g.mfma_stats()
            res.setdefault(nm, []).append((qb / t, (s1["second_pass_queries"] - s0["second_pass_queries"]) / 3, (s1["fallback_queries"] - s0["fallback_queries"]) / 3, g.last_dispatch()["kernel"]))
    nq = min(qb, 512)
    ke = torch.empty(nq * k, device=dev, dtype=torch.int64)
    ge_.search_topk_keys_dev(q.data_ptr(), nq, k, ke.data_ptr(), stream=st.cuda_stream); st.synchronize()
    for nm in gs:
        best = max(r[0] for r in res[nm])
        print(f"d={d} k={k} qb={qb:6d} {nm:7s} {best:10.0f} q/s  second-pass/call {res[nm][-1][1]:.1f} exact/call {res[nm][-1][2]:.1f}  {res[nm][-1][3]}  "
              f"keys = exact scan's on {nq}: {bool(torch.equal(ke, keys[nm][:nq * k]))}  equal to the other form: {bool(torch.equal(keys['slots'], keys['sample']))}", flush=True)
