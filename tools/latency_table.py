#!/usr/bin/env python3
"""Per-call latency of the host-pointer entry points, one query per call -- what the reference's own harnesses see
when they keep calling recognize() / predict() once per test image (ImageTesting.cpp:459-466, ann.cpp:97-103).
usage: python tools/latency_table.py"""
import os
import sys
import time

import numpy as np
import torch   # before the library: both bring a HIP runtime, torch's has to initialise first

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import synth  # noqa: E402


def timed(fn, reps):
    """Median per-call time in us (the interpreter's cyclic GC is off: with torch loaded one collection is ~40 ms)."""
    fn()
    fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e6


def main():
    import gc
    gc.disable()
    fir = ge.load_package()
    print(f"{'gallery':>16s} {'call':34s} {'us/call':>9s}   (median)")
    for n, d in ((3030, 1536), (100_000, 512), (1_000_000, 512)):
        rng = np.random.default_rng(n)
        rows = rng.random((n, d), dtype=np.float32)
        rows /= np.linalg.norm(rows, axis=1, keepdims=True)
        cls = synth.make_labels(n, 101)
        q = rows[:64] * np.float32(0.9) + rows[64:128] * np.float32(0.1)
        g = fir.Gallery(rows, cls, 0, 0)
        reps = 200 if n <= 100_000 else 30
        tag = f"{n}x{d}"
        i = [0]

        def one(fn):
            def call():
                i[0] = (i[0] + 1) % 64
                return fn(q[i[0]:i[0] + 1])
            return call
        print(f"{tag:>16s} {'search_top1, 1 query':34s} {timed(one(lambda v: g.search_top1(v)), reps):9.1f}")
        qd_ = torch.from_numpy(q).cuda()
        kd_ = torch.empty(64, dtype=torch.int64, device="cuda")
        st_ = torch.cuda.Stream()

        def dev_call(nq):
            def call():
                g.search_top1_keys_dev(qd_.data_ptr(), nq, kd_.data_ptr(), stream=st_.cuda_stream)
                st_.synchronize()
            return call
        print(f"{tag:>16s} {'keys_dev + sync, 1 query':34s} {timed(dev_call(1), reps):9.1f}")
        print(f"{tag:>16s} {'keys_dev + sync, 64 queries':34s} {timed(dev_call(64), max(reps // 4, 5)):9.1f}")
        print(f"{tag:>16s} {'search_top1, 8 queries':34s} {timed(lambda: g.search_top1(q[:8]), reps):9.1f}")
        print(f"{tag:>16s} {'search_top1, 64 queries':34s} {timed(lambda: g.search_top1(q), max(reps // 4, 5)):9.1f}")
        print(f"{tag:>16s} {'search_topk(5), 1 query':34s} {timed(one(lambda v: g.search_topk(v, 5)), reps):9.1f}")
        if d >= 256 and n <= 100_000:
            print(f"{tag:>16s} {'twd_conventional(post), 1 query':34s} {timed(one(lambda v: g.twd_conventional(v, 101, 0, 0.24, 64)), reps):9.1f}")
            print(f"{tag:>16s} {'twd_proposed(32), 1 query':34s} {timed(one(lambda v: g.twd_proposed(v, 32, 0.7)), reps):9.1f}")
            print(f"{tag:>16s} {'twd_conventional(post), 64 queries':34s} {timed(lambda: g.twd_conventional(q, 101, 0, 0.24, 64), max(reps // 8, 5)):9.1f}")
            print(f"{tag:>16s} {'twd_proposed(32), 64 queries':34s} {timed(lambda: g.twd_proposed(q, 32, 0.7), max(reps // 8, 5)):9.1f}")
        g.close()
    # classification.cpp side at its own scale: 3030 x 256 float64
    n, d = 3030, 256
    rng = np.random.default_rng(5)
    x = rng.random((n, d))
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    lab = np.sort(synth.make_labels(n, 101))
    avg, sd = x.mean(0), x.std(0)
    m = fir.ClsModel(x, lab, 101, avg, 0)
    qd = x[:64] * 0.9 + x[64:128] * 0.1
    i = [0]

    def onec(fn):
        def call():
            i[0] = (i[0] + 1) % 64
            return fn(qd[i[0]:i[0] + 1])
        return call
    tag = f"{n}x{d} f64"
    print(f"{tag:>16s} {'knn_predict(1), 1 query':34s} {timed(onec(lambda v: m.knn_predict(v, 1)), 200):9.1f}")
    print(f"{tag:>16s} {'pnn_predict, 1 query':34s} {timed(onec(lambda v: m.pnn_predict(v)), 200):9.1f}")
    print(f"{tag:>16s} {'pnn_predict_seq, 1 query':34s} {timed(onec(lambda v: m.pnn_predict_seq(v)), 200):9.1f}")
    m.close()
    f = fir.Fpnn(x, lab, 101, avg, sd, 1.0, 0)
    print(f"{tag:>16s} {'fpnn predict, 1 query':34s} {timed(onec(lambda v: f.predict(v)), 200):9.1f}")
    print(f"{tag:>16s} {'fpnn predict_seq, 1 query':34s} {timed(onec(lambda v: f.predict_seq(v)), 200):9.1f}")
    print(f"{tag:>16s} {'fpnn predict, 64 queries':34s} {timed(lambda: f.predict(qd), 50):9.1f}")
    f.close()


if __name__ == "__main__":
    main()
