#!/usr/bin/env python3
"""The reference's own experiment size (testRecognitionMethod, ImageTesting.cpp:439-501: Caltech-101-like, 30 gallery images
per class, the rest are queries, D = 1536): wall time of the recognition loop with the reference's CPU code (one thread,
as the reference runs it) against this library called the same way (one query per call) and with one batched call.
usage: python tools/reference_scale.py"""
import gc
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import oracle_lib  # noqa: E402


def main():
    gc.disable()
    fir = ge.load_package()
    classes, per_class_gallery, per_class_test, d = 101, 30, 50, 1536
    rng = np.random.default_rng(11)
    centers = rng.random((classes, d), dtype=np.float32)

    def draw(k):
        x = np.repeat(centers, k, axis=0) * np.float32(0.6) + rng.random((classes * k, d), dtype=np.float32)
        x[np.abs(x) < 1e-4] = 0
        return (x / np.linalg.norm(x, axis=1, keepdims=True)).astype(np.float32)

    gal, tst = draw(per_class_gallery), draw(per_class_test)
    gcls = np.repeat(np.arange(classes, dtype=np.int32), per_class_gallery)
    tcls = np.repeat(np.arange(classes, dtype=np.int32), per_class_test)
    n, nq = gal.shape[0], tst.shape[0]
    print(f"gallery {n} x {d}, {nq} queries, {classes} classes")

    # the reference, one thread, on a sample of the queries
    sample = np.arange(0, nq, max(nq // 100, 1))
    ref_rows = None
    if oracle_lib.have_ref():
        db = oracle_lib.load_ref("l2").db(gal, gcls, 0)
        t0 = time.perf_counter()
        ref_rows = [db.recognize_image_bf(tst[i], d) for i in sample]
        t_ref = (time.perf_counter() - t0) / len(sample)
        print(f"reference recognize_image_bf, 1 thread : {t_ref * 1e3:8.3f} ms per query  -> {t_ref * nq:8.2f} s for the test set")

    g = fir.Gallery(gal, gcls, 0, 0)
    g.search_top1(tst[:1])
    t0 = time.perf_counter()
    one = np.array([g.search_top1(tst[i:i + 1])[0][0] for i in range(nq)])
    t_one = time.perf_counter() - t0
    t0 = time.perf_counter()
    idx, _ = g.search_top1(tst)
    t_batch = time.perf_counter() - t0
    assert np.array_equal(one, idx)
    if ref_rows is not None:
        assert list(idx[sample]) == ref_rows, "same rows as the reference"
    acc = float(np.mean(gcls[idx] == tcls))
    print(f"this library, one query per call      : {t_one / nq * 1e6:8.1f} us per query  -> {t_one:8.3f} s for the test set")
    print(f"this library, one batched call        : {t_batch / nq * 1e6:8.2f} us per query  -> {t_batch * 1e3:8.2f} ms for the test set")
    print(f"accuracy {acc:.4f} (identical rows on the {len(sample)} sampled queries)")
    for name, fn in (("ConventionalTWD(posteriors, 0.24)", lambda: g.twd_conventional(tst, classes, 0, 0.24, 64)),
                     ("ProposedTWD(32, 0.7)", lambda: g.twd_proposed(tst, 32, 0.7))):
        fn()
        t0 = time.perf_counter()
        out = fn()
        t = time.perf_counter() - t0
        print(f"{name:38s}: {t / nq * 1e6:8.2f} us per query (batched), accuracy {float(np.mean(out[0] == tcls)):.4f}, "
              f"unreliable {float(np.mean(out[1] != 0)):.3f}")
    g.close()


if __name__ == "__main__":
    main()
