"""GPU parity of the three-way-decision classifiers (qt_cpp/ImageTesting.cpp:74-288): decisions
(class and reliable/unreliable) identical to the REAL reference's (tests/golden) and to the oracle."""
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))


def test_reference_decisions_reproduced(fir):
    rows, cls, q, ncls = gc.twd_case()
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for (typ, th) in gc.TWD_CONVENTIONAL:
            c, u = g.twd_conventional(q, ncls, typ, th, 64)
            assert np.array_equal(c, GOLD[f"twd_conv/{typ}_{th}/class"]), (typ, th)
            assert np.array_equal(u, GOLD[f"twd_conv/{typ}_{th}/unreliable"]), (typ, th)
        for (fc, th) in gc.TWD_PROPOSED:
            c, u, _ = g.twd_proposed(q, fc, th)
            assert np.array_equal(c, GOLD[f"twd_prop/{fc}_{th}/class"]), (fc, th)
            assert np.array_equal(u, GOLD[f"twd_prop/{fc}_{th}/unreliable"]), (fc, th)


@pytest.mark.parametrize("seed,n,ncls", [(5, 808, 101), (6, 3000, 37), (7, 70, 5), (8, 5000, 257)])
def test_matches_oracle_on_fresh_data(fir, oracle, seed, n, ncls):
    rows, cls, q, _ = gc.twd_case(seed=seed, n=n, d=280, n_classes=ncls)
    q = np.concatenate([q, q[:5] * np.float32(0.5) + rows[:5] * np.float32(0.5)])   # 17 queries: two internal batches + a ragged one
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for (typ, th) in gc.TWD_CONVENTIONAL + [(1, 0.0), (2, 0.5)]:
            c, u = g.twd_conventional(q, ncls, typ, th, 64)
            exp = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64) for qi in q]
            assert list(c) == [e[0] for e in exp], (typ, th)
            assert list(u) == [e[1] for e in exp], (typ, th)
        for (fc, th) in gc.TWD_PROPOSED + [(128, 0.7), (16, 0.5)]:
            c, u, k = g.twd_proposed(q, fc, th)
            exp = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in q]
            assert list(c) == [e[0] for e in exp], (fc, th)
            assert list(u) == [e[1] for e in exp], (fc, th)
            assert list(k) == [e[2] for e in exp], (fc, th)


@pytest.mark.parametrize("n,ncls,class_major", [(3840, 12, True), (3841, 12, False), (20000, 40, True), (11521, 7, False), (15361, 9, False), (60000, 101, True)])
def test_span_boundaries_and_class_major_galleries(fir, oracle, n, ncls, class_major):
    """The first-stage kernel takes the rows 3840 at a time and folds runs of equal labels before touching the class
    posteriors: galleries of exactly one span, one row more, several spans; labels class-major (the reference's
    order, ImageTesting.cpp:446) and interleaved."""
    rows, cls, q, _ = gc.twd_case(seed=21 + n % 7, n=n, d=256, n_classes=ncls)
    if class_major:
        cls = np.sort(cls)
    q = q[:6]
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for (typ, th) in gc.TWD_CONVENTIONAL:
            c, u = g.twd_conventional(q, ncls, typ, th, 64)
            exp = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64) for qi in q]
            assert list(c) == [e[0] for e in exp] and list(u) == [e[1] for e in exp], (typ, th)
        for (fc, th) in gc.TWD_PROPOSED:
            c, u, k = g.twd_proposed(q, fc, th)
            exp = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in q]
            assert list(c) == [e[0] for e in exp] and list(u) == [e[1] for e in exp] and list(k) == [e[2] for e in exp], (fc, th)


def test_second_best_is_order_dependent(fir, oracle):
    """secondBestDist follows the scan order (ImageTesting.cpp:123-125), not 'best of the other classes':
    rows are arranged so that the two differ."""
    rows, cls, q, ncls = gc.twd_case(seed=11, n=600, d=256, n_classes=6)
    order = np.argsort(oracle.all_distances(rows, q[0], 0, 64, 0))[::-1].copy()    # descending distance: every row is a new best
    rows, cls = rows[order], cls[order]
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for typ, th in ((1, 1e-4), (1, 1e-3), (2, 0.9), (2, 0.99)):
            c, u = g.twd_conventional(q[:3], ncls, typ, th, 64)
            exp = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64) for qi in q[:3]]
            assert list(c) == [e[0] for e in exp] and list(u) == [e[1] for e in exp]


def test_argument_errors(fir):
    rows, cls, q, ncls = gc.twd_case(seed=12, n=100, d=256, n_classes=6)
    with fir.Gallery(rows, None, gc.L2, 0) as g:
        with pytest.raises(fir.FirError):
            g.twd_conventional(q, ncls, 0, 0.24)            # no labels
    with fir.Gallery(rows[:, :128].copy(), cls, gc.L2, 0) as g:
        with pytest.raises(fir.FirError):
            g.twd_proposed(q[:, :128].copy(), 32, 0.7)      # needs 256 features
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        with pytest.raises(fir.FirError):
            g.twd_conventional(q, 4, 0, 0.24)               # top-5 posteriors need >= 5 classes
        with pytest.raises(fir.FirError):
            g.twd_proposed(q, 48, 0.7)                      # 48 does not divide 256
