#!/usr/bin/env python3
"""K3 float64 scan forms (FIR_CLS_FORM = NT,R,U,BLOCK,WPS) over a 1M x 512 training set, interleaved rounds in ONE process:
kernel ms per launch (HIP events in the library), kNN-1 queries/s of the whole call, classes equal to the default form's.
usage: python tools/k3_sweep.py [--queries 64] [--rounds 3] [--forms "2,1,8,256,2;2,2,4,256,2;..."]"""
import argparse, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--forms", default="default;2,1,8,256,2;2,2,4,256,2;2,2,2,256,2;2,2,8,256,2;2,2,4,512,2;3,1,8,512,2;3,2,4,512,2;4,1,8,512,2;4,1,4,512,2;4,2,2,512,2;2,1,4,512,4;2,4,2,256,2")
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    n, d, ncls, qb = 1_000_000, 512, 1000, a.queries
    g = torch.Generator(device=dev); g.manual_seed(31337)
    centres = torch.rand((ncls, d), generator=g, device=dev, dtype=torch.float64)
    tcls = (torch.arange(n, device=dev) * ncls // n).to(torch.int64)
    tr = torch.empty((n, d), device=dev, dtype=torch.float64)
    for lo in range(0, n, 125_000):
        hi = lo + 125_000
        tr[lo:hi] = centres[tcls[lo:hi]] + 0.004 * torch.randn((hi - lo, d), generator=g, device=dev, dtype=torch.float64)
    avg = tr.mean(dim=0).cpu().numpy()
    pick = torch.randint(0, ncls, (qb,), generator=g, device=dev)
    q = (centres[pick] + 0.004 * torch.randn((qb, d), generator=g, device=dev, dtype=torch.float64)).cpu().numpy()
    torch.cuda.synchronize()
    m = fir.ClsModel(None, tcls.to(torch.int32).cpu().numpy(), ncls, avg, dev.index, dev_ptr=tr.data_ptr(), nt=n, d=d)
    del tr
    torch.cuda.empty_cache()
    m.profile_enable(True)
    forms = a.forms.split(";")
    ref = None
    res = {f: {"ms": [], "qps": []} for f in forms}
    names = {}
    for rnd in range(a.rounds):
        for f in forms:
            if f == "default":
                os.environ.pop("FIR_CLS_FORM", None)
            else:
                os.environ["FIR_CLS_FORM"] = f
            cls, sc = m.pnn_predict(q)            # warm
            m.profile_read()
            t0 = time.perf_counter()
            for _ in range(2):
                k1 = m.knn_predict(q, 1)
            dt = (time.perf_counter() - t0) / 2
            ms, nbytes, kname = m.profile_read()
            names[f] = kname
            if ref is None:
                ref = (cls.copy(), sc.copy(), k1.copy())
            same = bool(np.array_equal(cls, ref[0]) and np.array_equal(sc.view(np.uint64), ref[1].view(np.uint64)) and np.array_equal(k1, ref[2]))
            res[f]["ms"].append(float(np.mean(ms)) if len(ms) else float("nan"))
            res[f]["qps"].append(qb / dt)
            res[f]["same"] = res[f].get("same", True) and same
            res[f]["launches"] = len(ms) // 2 if len(ms) else 0
    winst = (n / 64.0) * d * qb * 3.0
    print(f"{'form NT,R,U,BLOCK,WPS':24s} {'kernel':44s} {'launches/call':>13s} {'ms/call (kernels)':>18s} {'f64 issue frac':>14s} {'kNN-1 q/s':>10s}  bits equal")
    for f in forms:
        r = res[f]
        msc = min(r["ms"]) * r["launches"]
        print(f"{f:24s} {names[f][:44]:44s} {r['launches']:13d} {msc:18.3f} {winst / (msc * 1e-3) / 1e9 / 614.4:14.3f} {max(r['qps']):10.0f}  {r['same']}", flush=True)
    m.close()

if __name__ == "__main__":
    main()
