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


class _FusedMode:
    """FIR_TWD_FUSED for the calls inside: 0 = the launch-per-chunk forms, 1 = automatic (up to 8 queries per call go through
    k_twd_prop_fused), 2 = the fused form whatever the batch (8 queries per launch)."""

    def __init__(self, mode):
        self.mode = str(mode)

    def __enter__(self):
        self.old = os.environ.get("FIR_TWD_FUSED")
        os.environ["FIR_TWD_FUSED"] = self.mode

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("FIR_TWD_FUSED", None)
        else:
            os.environ["FIR_TWD_FUSED"] = self.old


def _proposed_all_forms(g, q, fc, th):
    out = []
    for mode in (0, 1, 2):
        with _FusedMode(mode):
            out.append(tuple(np.asarray(x).tolist() for x in g.twd_proposed(q, fc, th)))
    return out


@pytest.mark.parametrize("n,ncls,metric", [(70, 5, gc.L2), (3000, 37, gc.L2), (8192, 11, gc.L2), (20001, 40, gc.L2), (100000, 101, gc.L2),
                                           (5000, 23, gc.CHI2), (40000, 64, gc.CHI2)])
def test_proposed_twd_as_one_launch_matches_the_oracle_and_the_other_forms(fir, oracle, n, ncls, metric):
    """k_twd_prop_fused (the chunk distances stay in registers, the workgroups of a query meet once per chunk) against the
    oracle and against the launch-per-chunk forms: class, unreliable flag and the number of chunks used, for thresholds
    that stop after one chunk, after a few, and never (th = 0.3: every chunk; th = 1.5: even the best row is pruned)."""
    rows, cls, q, _ = gc.twd_case(seed=31 + n % 11, n=n, d=256, n_classes=ncls)
    if metric == gc.CHI2:
        rows = np.abs(rows) + np.float32(1e-3)
        q = np.abs(q) + np.float32(1e-3)
    q = np.concatenate([q[:5], q[5:8] * np.float32(0.05) + rows[[1, n // 2, n - 1]] * np.float32(0.95)])    # 8 queries, three of them next to a row
    with fir.Gallery(rows, cls, metric, 0) as g:
        for (fc, th) in [(32, 0.7), (64, 0.95), (16, 0.3), (128, 0.7), (32, 1.5), (4, 0.9)]:
            if n >= 40000 and fc == 4:
                continue                                              # (64 oracle chunks over 100k rows: seconds per query, covered below 40k)
            exp = [oracle.twd_proposed(rows, cls, qi, fc, th, metric) for qi in q]
            exp = ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp])
            for form, got in zip(("per-chunk", "auto", "fused"), _proposed_all_forms(g, q, fc, th)):
                assert got == exp, (form, fc, th)
            with _FusedMode(1):
                c1, u1, k1 = g.twd_proposed(q[6:7], fc, th)             # the one-query call (all CUs on one query)
            assert (int(c1[0]), int(u1[0]), int(k1[0])) == (exp[0][6], exp[1][6], exp[2][6]), (fc, th)


def test_proposed_twd_fused_ties_duplicates_and_rows_nothing_qualifies(fir, oracle):
    """Equal sums in different workgroups (duplicated rows far apart: the FIRST row is the best one and its class decides),
    a gallery whose every distance is above the reference's 100000 start value (bestInd stays -1: class -1 after one chunk),
    NaN rows (never the best, never pruned, counted as survivors of their class)."""
    rows, cls, q, _ = gc.twd_case(seed=44, n=30000, d=256, n_classes=50)
    q = q[:4].copy()
    dup = rows[17].copy()
    for r in (17, 9000, 9001, 29999):
        rows[r] = dup
    cls[17], cls[9000], cls[9001], cls[29999] = 3, 4, 3, 5
    q[0] = dup
    q[1] = dup * np.float32(1.001)
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for (fc, th) in [(32, 0.7), (64, 0.999), (32, 1.0)]:
            exp = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in q]
            exp = ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp])
            for form, got in zip(("per-chunk", "auto", "fused"), _proposed_all_forms(g, q, fc, th)):
                assert got == exp, (form, fc, th)
    rows2 = rows.copy()
    rows2[5] = np.nan
    rows2[12345, 40] = np.nan
    with fir.Gallery(rows2, cls, gc.L2, 0) as g:
        exp = [oracle.twd_proposed(rows2, cls, qi, 32, 0.7) for qi in q]
        exp = ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp])
        for form, got in zip(("per-chunk", "auto", "fused"), _proposed_all_forms(g, q, 32, 0.7)):
            assert got == exp, form
    far = np.full((700, 256), 3.0e4, np.float32)                       # every chunk distance is 9e8 > 100000
    with fir.Gallery(far, np.arange(700, dtype=np.int32) % 7, gc.L2, 0) as g:
        forms = _proposed_all_forms(g, np.zeros((2, 256), np.float32), 32, 0.7)
        assert forms[0] == forms[1] == forms[2] == ([-1, -1], [0, 0], [1, 1])


def test_proposed_twd_fused_many_calls_in_a_row(fir, oracle):
    """The two state blocks of the fused form alternate from call to call and each call clears the other one: a few hundred calls
    with changing chunk counts, one and several queries, against the oracle."""
    rows, cls, q, _ = gc.twd_case(seed=52, n=12000, d=256, n_classes=30)
    rng = np.random.default_rng(5)
    with fir.Gallery(rows, cls, gc.L2, 0) as g, _FusedMode(1):
        for it in range(150):
            fc = (8, 32, 64, 128)[it % 4]
            th = (0.7, 0.95, 0.4)[it % 3]
            nq = 1 + it % 5
            qq = q[rng.integers(0, len(q), nq)] * np.float32(0.5) + rows[rng.integers(0, len(rows), nq)] * np.float32(0.5)
            c, u, k = g.twd_proposed(qq, fc, th)
            exp = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in qq]
            assert (list(c), list(u), list(k)) == ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp]), (it, fc, th)


def test_proposed_twd_fused_beside_a_busy_device(fir, oracle):
    """The workgroups of k_twd_prop_fused wait for each other: run it while another host thread keeps the device busy with
    large batched searches on another gallery (its own stream). Every call must come back with the oracle's answer -- through
    the one-launch form when its workgroups get their CUs in time, through the launch-per-chunk form when they do not."""
    import threading

    rows, cls, q, _ = gc.twd_case(seed=63, n=40000, d=256, n_classes=50)
    rng = np.random.default_rng(9)
    big = rng.random((200000, 128), dtype=np.float32)
    bq = rng.random((4096, 128), dtype=np.float32)
    stop = threading.Event()
    errors = []

    def hammer():
        try:
            with fir.Gallery(big, None, gc.L2, 0) as gb:
                while not stop.is_set():
                    gb.search_top1(bq)
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append(e)

    t = threading.Thread(target=hammer)
    t.start()
    try:
        with fir.Gallery(rows, cls, gc.L2, 0) as g, _FusedMode(1):
            for it in range(60):
                nq = 1 + it % 3
                qq = q[rng.integers(0, len(q), nq)] * np.float32(0.6) + rows[rng.integers(0, len(rows), nq)] * np.float32(0.4)
                fc, th = ((32, 0.7), (64, 0.9), (16, 0.5))[it % 3]
                c, u, k = g.twd_proposed(qq, fc, th)
                exp = [oracle.twd_proposed(rows, cls, qi, fc, th) for qi in qq]
                assert (list(c), list(u), list(k)) == ([e[0] for e in exp], [e[1] for e in exp], [e[2] for e in exp]), (it, fc, th)
    finally:
        stop.set()
        t.join(timeout=60)
    assert not errors, errors


def _conventional_all_forms(g, q, ncls, typ, th, fc=64):
    out = []
    for mode in (0, 1, 2):
        with _FusedMode(mode):
            out.append(tuple(np.asarray(x).tolist() for x in g.twd_conventional(q, ncls, typ, th, fc)))
    return out


@pytest.mark.parametrize("n,ncls,metric,class_major", [(70, 5, gc.L2, False), (3000, 37, gc.L2, True), (8192, 11, gc.L2, False), (20001, 40, gc.L2, True),
                                                        (100000, 101, gc.L2, True), (5000, 23, gc.CHI2, False), (40000, 64, gc.CHI2, True)])
def test_conventional_twd_as_one_launch_matches_the_oracle_and_the_other_forms(fir, oracle, n, ncls, metric, class_major):
    """k_twd_conv_fused (both partial distances stay in registers, two meetings, secondBestDist from the workgroups' last local records)
    against the oracle and the launch-per-stage forms: every type, thresholds on both sides of the reliability test, class-major
    (the reference's order) and interleaved labels, 8 queries per call and one."""
    rows, cls, q, _ = gc.twd_case(seed=41 + n % 13, n=n, d=256, n_classes=ncls)
    if class_major:
        cls = np.sort(cls)
    if metric == gc.CHI2:
        rows = np.abs(rows) + np.float32(1e-3)
        q = np.abs(q) + np.float32(1e-3)
    q = np.concatenate([q[:5], q[5:8] * np.float32(0.05) + rows[[1, n // 2, n - 1]] * np.float32(0.95)])
    with fir.Gallery(rows, cls, metric, 0) as g:
        for (typ, th) in gc.TWD_CONVENTIONAL + [(1, 0.0), (2, 0.5), (0, 0.9)]:
            for fc in (64, 32) if n <= 20001 else (64,):
                exp = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, fc, metric) for qi in q]
                exp = ([e[0] for e in exp], [e[1] for e in exp])
                for form, got in zip(("per-stage", "auto", "fused"), _conventional_all_forms(g, q, ncls, typ, th, fc)):
                    assert got == exp, (form, typ, th, fc)
                with _FusedMode(1):
                    c1, u1 = g.twd_conventional(q[6:7], ncls, typ, th, fc)
                assert (int(c1[0]), int(u1[0])) == (exp[0][6], exp[1][6]), (typ, th, fc)


def test_conventional_twd_fused_second_best_follows_the_scan_order(fir, oracle):
    """secondBestDist is 'bestDist at the last class change of the record walk' (ImageTesting.cpp:123-125), not 'the best of the other
    classes': rows in descending distance (every row is a new record, spread over many workgroups), duplicated rows far apart (equal
    distances: the first one is the record), NaN rows, and a gallery in which nothing is below the 100000 start value."""
    rows, cls, q, ncls = gc.twd_case(seed=17, n=30000, d=256, n_classes=9)
    order = np.argsort(oracle.all_distances(rows, q[0], 0, 64, 0))[::-1].copy()
    rows, cls = rows[order], cls[order]
    rows[20000] = rows[100]
    rows[20001] = rows[29999]
    q = q[:4].copy()
    q[1] = rows[29999] * np.float32(0.999)
    rows2 = rows.copy()
    rows2[7] = np.nan
    rows2[25000, 3] = np.nan
    for rr in (rows, rows2):
        with fir.Gallery(rr, cls, gc.L2, 0) as g:
            for typ, th in ((1, 1e-4), (1, 1e-3), (1, 1e-6), (2, 0.9), (2, 0.99), (2, 0.9999), (0, 0.24)):
                exp = [oracle.twd_conventional(rr, cls, qi, ncls, typ, th, 64) for qi in q]
                exp = ([e[0] for e in exp], [e[1] for e in exp])
                for form, got in zip(("per-stage", "auto", "fused"), _conventional_all_forms(g, q, ncls, typ, th)):
                    assert got == exp, (form, typ, th)
    far = np.full((700, 256), 3.0e4, np.float32)
    with fir.Gallery(far, np.arange(700, dtype=np.int32) % 7, gc.L2, 0) as g:
        for typ, th in ((0, 0.24), (1, 0.003), (2, 0.7)):
            forms = _conventional_all_forms(g, np.zeros((2, 256), np.float32), 7, typ, th)
            assert forms[0] == forms[1] == forms[2], (typ, forms)


def test_conventional_twd_fused_many_calls_in_a_row(fir, oracle):
    """The fused form leaves its state words zero behind every call (atomic exchanges by the deciding workgroup): a few hundred
    calls of changing type, batch and threshold against the oracle."""
    rows, cls, q, ncls = gc.twd_case(seed=58, n=12000, d=256, n_classes=30)
    rng = np.random.default_rng(6)
    with fir.Gallery(rows, cls, gc.L2, 0) as g, _FusedMode(1):
        for it in range(150):
            typ, th = gc.TWD_CONVENTIONAL[it % len(gc.TWD_CONVENTIONAL)]
            nq = 1 + it % 5
            qq = q[rng.integers(0, len(q), nq)] * np.float32(0.5) + rows[rng.integers(0, len(rows), nq)] * np.float32(0.5)
            c, u = g.twd_conventional(qq, ncls, typ, th, 64)
            exp = [oracle.twd_conventional(rows, cls, qi, ncls, typ, th, 64) for qi in qq]
            assert (list(c), list(u)) == ([e[0] for e in exp], [e[1] for e in exp]), (it, typ, th)
