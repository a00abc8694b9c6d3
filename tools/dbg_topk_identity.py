import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
fir = ge.load_package()
import test_gpu_gemm as T
dev = torch.device("cuda", 0)
os.environ["FIR_GEMM_ADAPTIVE"] = "0"; os.environ["FIR_GEMM_SAMPLE_DIV"] = "1000000"
def run(k):
    n_ids, per, d, qb = 25000, 40, 128, 384
    rows, centres = T._identity_gallery(n_ids, per, d, 77, dev)
    gq = torch.Generator(device=dev); gq.manual_seed(78)
    who = torch.randint(0, n_ids, (qb,), generator=gq, device=dev)
    q = centres[who] * (1 + 0.05 * (torch.rand((qb, d), generator=gq, device=dev) - 0.5))
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    whon = who.cpu().numpy()
    ke = torch.empty(qb * k, dtype=torch.int64, device=dev)
    km = torch.empty(qb * k, dtype=torch.int64, device=dev)
    with fir.Gallery(dev_ptr=rows.data_ptr(), n=n_ids * per, d=d, metric=0, device=0) as g:
        g.set_large_batch_mfma(0)
        if k == 1: g.search_top1_keys_dev(q.data_ptr(), qb, ke.data_ptr())
        else: g.search_topk_keys_dev(q.data_ptr(), qb, k, ke.data_ptr())
        torch.cuda.synchronize()
        idx, dd = fir.keys_unpack(ke.cpu().numpy().view(np.uint64))
        bad = np.nonzero(idx.reshape(qb, k)[:, 0] // per != whon)[0]
        print(k, "exact: wrong-identity queries", bad[:10], len(bad), g.last_dispatch()["kernel"], "who0", whon[0], "rows ptr", hex(rows.data_ptr()))
        with fir.GemmSearch(g, 2) as m:
            if k == 1: m.search_top1_keys_dev(q.data_ptr(), qb, km.data_ptr())
            else: m.search_topk_keys_dev(q.data_ptr(), qb, k, km.data_ptr())
            torch.cuda.synchronize()
            print(k, "stats", m.stats())
    idx2, dd2 = fir.keys_unpack(km.cpu().numpy().view(np.uint64))
    bad2 = np.nonzero(idx2.reshape(qb, k)[:, 0] // per != whon)[0]
    print(k, "gemm: wrong-identity queries", bad2[:10], len(bad2), "equal", bool(torch.equal(ke, km)))
    if len(bad2):
        i = bad2[0]
        print("q", i, "who", whon[i], "gemm ids", idx2.reshape(qb, k)[i] // per, dd2.reshape(qb, k)[i], "exact ids", idx.reshape(qb, k)[i] // per, dd.reshape(qb, k)[i])
        # is the row the library names really nearer? (torch f32)
        r = int(idx2.reshape(qb, k)[i, 0])
        print("torch dist to named row", float(((q[i] - rows[r]) ** 2).sum() / d), "to own identity's best", float((((q[i][None] - rows[whon[i]*per:(whon[i]+1)*per]) ** 2).sum(1) / d).min()))
for k in (1, 5, 5):
    run(k)
