import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as ge
fir = ge.load_package()
dev = torch.device("cuda", 0)
n, d = 1_000_000, 512
x = torch.rand((n, d), device=dev); x = x / x.norm(dim=1, keepdim=True)
g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
st = torch.cuda.Stream()
for qb in (8, 32, 256, 1024, 4096, 16384):
    q = torch.rand((qb, d), device=dev); q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    keys = torch.empty(qb * 5, device=dev, dtype=torch.int64)
    k1 = torch.empty(qb, device=dev, dtype=torch.int64)
    def f(): g.search_topk_keys_dev(q.data_ptr(), qb, 5, keys.data_ptr(), stream=st.cuda_stream); st.synchronize()
    def f1(): g.search_top1_keys_dev(q.data_ptr(), qb, k1.data_ptr(), stream=st.cuda_stream); st.synchronize()
    for fn, nm in ((f, "top-5"), (f1, "top-1")):
        fn(); fn()
        t0 = time.perf_counter()
        for _ in range(5): fn()
        t = (time.perf_counter() - t0) / 5
        print(f"qb={qb:5d} {nm}: {t*1e3:8.3f} ms  {qb/t:9.0f} q/s")
    assert torch.equal(keys.view(qb, 5)[:, 0], k1)
    if qb >= 128:                 # the exact top-K scan beside the default (matrix-core) dispatch
        g.set_large_batch_mfma(0)
        ke = torch.empty(qb * 5, device=dev, dtype=torch.int64)
        nq = min(qb, 1024)
        t0 = time.perf_counter()
        g.search_topk_keys_dev(q.data_ptr(), nq, 5, ke.data_ptr(), stream=st.cuda_stream); st.synchronize()
        t = time.perf_counter() - t0
        g.set_large_batch_mfma(-1)
        print(f"qb={nq:5d} top-5 exact scan: {t*1e3:8.3f} ms  {nq/t:9.0f} q/s   identical {torch.equal(ke[:nq*5], keys[:nq*5])}")
