#!/usr/bin/env python3
"""A/B of matrix-core path variants in ONE process, interleaved rounds (guide rule 24): every variant is a fir_gemm handle
created under its own environment settings (FIR_GEMM_* are read at create time) on the same gallery; per round every variant
answers the same queries. Reports wall queries/s (median, best) and the full-pass kernel time from the library's HIP events.
usage: python tools/mfma_ab.py --dim 512 --qb 32768 --variants "base:;x16:FIR_GEMM_MFMA16=1;x16s:FIR_GEMM_MFMA16=1,FIR_GEMM_STAGGER=1"
"""
import argparse
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=512)
    ap.add_argument("--qb", type=int, default=32768)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--check", type=int, default=2048, help="queries checked against the exact scan per variant")
    ap.add_argument("--variants", default="base:;x16:FIR_GEMM_MFMA16=1")
    ap.add_argument("--identities", type=int, default=0, help="> 0: a class-ordered clustered gallery, rows = identity centre * (1 + spread * noise), identity = row * identities // rows")
    ap.add_argument("--spread", type=float, default=0.05)
    a = ap.parse_args()
    fir = ge.load_package()
    dev = torch.device("cuda", 0)
    n, d, qb = a.rows, a.dim, a.qb
    torch.manual_seed(13)
    if a.identities > 0:
        centres = torch.rand((a.identities, d), device=dev)
        ident = (torch.arange(n, device=dev) * a.identities) // n          # class-ordered, like the reference's galleries
        x = centres[ident] * (1 + a.spread * (torch.rand((n, d), device=dev) - 0.5))
        del centres, ident
    else:
        x = torch.rand((n, d), device=dev)
    x = x / x.norm(dim=1, keepdim=True)
    g = fir.Gallery(dev_ptr=x.data_ptr(), n=n, d=d, metric=0, device=0)
    fresh = torch.rand((qb, d), device=dev)
    pert = x[(torch.arange(qb, device=dev) * 977 + 11) % n] + (torch.rand((qb, d), device=dev) - 0.5) * 0.05 * x.mean()
    q = torch.where((torch.arange(qb, device=dev) % 2 == 0)[:, None], fresh, pert.clamp_min(0))
    q = (q / q.norm(dim=1, keepdim=True)).contiguous()
    del fresh, pert
    st = torch.cuda.Stream()
    nchk = min(a.check, qb)
    kref = torch.empty(max(nchk, 1), device=dev, dtype=torch.int64)[:nchk]
    if nchk > 0:
        with torch.cuda.stream(st):
            g.set_large_batch_mfma(0)
            g.search_top1_keys_dev(q.data_ptr(), nchk, kref.data_ptr(), stream=st.cuda_stream)
            g.set_large_batch_mfma(-1)
    torch.cuda.synchronize()
    variants = []
    for spec in a.variants.split(";"):
        name, _, envs = spec.partition(":")
        env = dict(kv.split("=") for kv in envs.split(",") if kv)
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        m = fir.GemmSearch(g, 2)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        variants.append((name, m))
    keys = torch.empty(qb, device=dev, dtype=torch.int64)
    res = {name: {"wall": [], "kern": [], "same": None, "fb": 0} for name, _ in variants}
    g.profile_enable(True)
    for rnd in range(a.rounds + 1):                      # round 0 is the warm-up and the check
        for name, m in variants:
            keys.zero_()
            g.profile_read()
            s0 = m.stats()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                m.search_top1_keys_dev(q.data_ptr(), qb, keys.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
            t = time.perf_counter() - t0
            ms, _ = g.profile_read()
            if rnd == 0:
                res[name]["same"] = bool(torch.equal(keys[:nchk], kref))
                res[name]["fb"] = m.stats()["fallback_queries"] - s0["fallback_queries"]
                res[name]["disp"] = g.last_dispatch()
                continue
            res[name]["wall"].append(t)
            res[name]["kern"].append(float(ms.sum()) / max(1, len(ms))); res[name].setdefault("kmin", []).append(float(ms.min()))
            res[name]["launches"] = len(ms)
    print(f"gallery {n} x {d}, {qb} queries per call, {a.rounds} interleaved rounds")
    for name, _ in variants:
        r = res[name]
        w = sorted(r["wall"])
        k = sorted(r["kern"])
        disp = r["disp"]
        fl = disp["flops_per_launch"]
        print(f"{name:10s} {qb / statistics.median(w):10.0f} q/s median ({qb / w[0]:10.0f} best)  kernel {statistics.median(k):7.4f} ms x {r['launches']} launches "
              f"(best {k[0]:7.4f}, fastest launch {min(r['kmin']):7.4f}) = {fl / statistics.median(k) / 1e9:7.1f} TFLOP/s  {disp['kernel']} vgprs={disp['vgprs']}  identical keys ({nchk}): {r['same']}  fallbacks: {r['fb']}",
              flush=True)
    for _, m in variants:
        m.close()
    g.close()


if __name__ == "__main__":
    main()
