"""GPU parity of the DirectedEnumeration pieces (qt_cpp/ann.cpp:270-507, PIVOT build): the pivot table and greedy
pivots of the constructor, the per-query likelihood update, and the candidate-row distance gather -- against the REAL
reference's outputs (tests/golden) and against the oracle on fresh data."""
import os

import numpy as np
import pytest

import golden_cases as gc
import synth

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_reference_pivot_table_reproduced(fir, oracle):
    rows, cls, _ = gc.dem_case()
    gp, gt, gth = GOLD["dem/pivots"], GOLD["dem/table"], GOLD["dem/threshold"]
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        piv, table, mo, built = g.dem_pivot_table(int(gp[0]), gp.size)
        assert built == gp.size and np.array_equal(piv, gp)
        assert np.array_equal(bits(table), bits(gt))
        assert bits(oracle.get_threshold(mo, 0.01)) == bits(gth)
        piv2, none, mo2, _ = g.dem_pivot_table(int(gp[0]), gp.size, want_table=False)      # one scratch row instead of the table
        assert none is None and np.array_equal(piv2, gp) and np.array_equal(bits(mo2), bits(mo))


def test_reference_walk_reproduced_from_device_pieces(fir, oracle):
    """recognize (ann.cpp:411-507) rebuilt from the device outputs -- pivot distances, likelihoods, candidate distances --
    gives the reference's row / bestDistance / found / distanceCalcCount."""
    rows, cls, queries = gc.dem_case()
    gp, gth = GOLD["dem/pivots"], GOLD["dem/threshold"]
    n = len(rows)
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        dem = fir.Dem(g, int(gp[0]), gp.size)
        assert (dem.n_pivots, dem.n_built, dem.n_used, dem.n) == (gp.size, gp.size, gp.size, n)
        piv, mo, table, order = dem.get()
        assert np.array_equal(piv, gp) and np.array_equal(bits(table), bits(GOLD["dem/table"]))
        pd, lik = dem.likelihoods(queries)
        for m in gc.DEM_IMAGE_COUNTS:
            M = m if 0 < m < n else n
            for i, q in enumerate(queries):
                want = tuple(GOLD[f"dem/recognize/{m}/{k}"][i] for k in ("row", "dist", "found", "calc"))
                best, row, calc, found = np.float32(np.finfo(np.float32).max), -1, 0, 0
                for k in range(dem.n_used):
                    calc += 1
                    if pd[i, k] < best:
                        best, row = pd[i, k], int(piv[k])
                        if best < gth:
                            found = 1
                            break
                if not found:
                    cand = order[dem.n_used:]
                    cand = cand[np.argsort(lik[i][cand], kind="stable")][: M - dem.n_used]
                    dist = g.rows_distances(q, cand)[0]
                    for k, dk in enumerate(dist):
                        calc += 1
                        if dk < best:
                            best, row = dk, int(cand[k])
                            if best < gth:
                                found = 1
                                break
                assert (row, bits(best), found, calc) == (want[0], bits(want[1]), want[2], want[3]), (m, i)
        dem.close()


@pytest.mark.parametrize("seed,n,d,ncls,npiv,metric", [(31, 2000, 512, 40, 30, gc.L2), (32, 777, 100, 7, 40, gc.L2),
                                                        (33, 1500, 256, 30, 12, gc.CHI2), (34, 64, 33, 4, 5, gc.L2)])
def test_matches_oracle_on_fresh_data(fir, oracle, seed, n, d, ncls, npiv, metric):
    rows = synth.make_gallery(seed, n, d, metric)
    cls = synth.make_labels(n, ncls)
    q, _ = synth.make_queries(seed, rows, 11, metric)
    first = (seed * 7919) % n
    epiv, etable, emo = oracle.dem_pivot_table(rows, cls, first, npiv, metric)
    with fir.Gallery(rows, cls, metric, 0) as g:
        piv, table, mo, built = g.dem_pivot_table(first, npiv)
        assert built == npiv and np.array_equal(piv, epiv)
        assert np.array_equal(bits(table), bits(etable)) and np.array_equal(bits(mo), bits(emo))
        dem = fir.Dem(g, first, npiv)
        used = min(npiv, 32)
        assert dem.n_used == used
        pd, lik = dem.likelihoods(q)                      # 11 queries: one full internal batch + a ragged one
        for i, qi in enumerate(q):
            # threshold 0 -> no early exit: the oracle's likelihoods after all kept pivots
            *_, elik = oracle.dem_recognize(rows, epiv[:used], etable[:used], 0.0, used, qi, metric, want_lik=True)
            assert np.array_equal(bits(lik[i]), bits(elik)), i
            assert np.array_equal(bits(pd[i]), bits([oracle.feature_distance(qi, rows[p], 0, d, metric) for p in epiv[:used]]))
        dem.close()


def test_index_bookkeeping_quirk_is_reproduced(fir, oracle):
    """The reference moves a pivot to the front of its index array with two plain writes (ann.cpp:431-432); when a pivot
    sits at a position below the number of pivots the array loses one row and holds another twice, so some rows are
    updated twice per pivot and some never. Row 0 is made an outlier so that the greedy choice picks it as the second
    pivot after first_pivot = 7, which forces that; the likelihoods still match."""
    n, d = 300, 64
    rows = synth.make_gallery(41, n, d, gc.L2)
    rows[0] = 0
    rows[0, 0] = 1
    cls = synth.make_labels(n, 10)
    q, _ = synth.make_queries(41, rows, 4, gc.L2)
    epiv, etable, _ = oracle.dem_pivot_table(rows, cls, 7, 5, gc.L2)
    assert epiv[1] == 0
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        dem = fir.Dem(g, 7, 5)
        piv, _, _, order = dem.get(want_table=False)
        assert np.array_equal(piv, epiv)
        li = np.arange(n)
        for s, p in enumerate(epiv):                      # the reference's two writes
            li[p] = li[s]
            li[s] = p
        assert np.array_equal(order, li)
        assert len(set(li.tolist())) < n                  # the quirk did happen: the array is no longer a permutation
        _, lik = dem.likelihoods(q)
        for i, qi in enumerate(q):
            *_, elik = oracle.dem_recognize(rows, epiv, etable, 0.0, 5, qi, gc.L2, want_lik=True)
            assert np.array_equal(bits(lik[i]), bits(elik)), i
        dem.close()


def test_rows_distances_is_the_scan_arithmetic(fir, oracle):
    for metric, d, (s, e) in ((gc.L2, 130, (0, 0)), (gc.L2, 256, (5, 77)), (gc.CHI2, 96, (0, 0))):
        rows = synth.make_gallery(51, 900, d, metric)
        q, _ = synth.make_queries(51, rows, 3, metric)
        pick = (synth.splitmix64(np.arange(3 * 50, dtype=np.uint64), 9) % np.uint64(900)).astype(np.int32).reshape(3, 50)
        pick[0, 0], pick[1, 1] = -1, 900                  # outside the gallery -> 100000
        with fir.Gallery(rows, None, metric, 0) as g:
            got = g.rows_distances(q, pick, s, e)
            full = g.range_distances(q, s, e)
        for i in range(3):
            for k in range(50):
                r = pick[i, k]
                exp = np.float32(100000.0) if r < 0 or r >= 900 else oracle.feature_distance(q[i], rows[r], s, e or d, metric)
                assert bits(got[i, k]) == bits(exp), (metric, i, k)
                if 0 <= r < 900:
                    assert bits(got[i, k]) == bits(full[i, r])


def test_degenerate_gallery_and_argument_errors(fir):
    rows = np.tile(synth.make_gallery(61, 1, 64, gc.L2), (50, 1))       # identical rows: no positive far sum after pivot 0
    cls = synth.make_labels(50, 5)
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        piv, _, mo, built = g.dem_pivot_table(3, 5)
        assert built == 1 and piv[0] == 3 and (piv[1:] == -1).all() and mo[0] == 0
        for bad in ((-1, 5), (50, 5), (0, 0)):
            with pytest.raises(fir.FirError):
                g.dem_pivot_table(*bad)
    with fir.Gallery(rows, None, gc.L2, 0) as g:
        with pytest.raises(fir.FirError):
            g.dem_pivot_table(0, 5)                                      # no class labels
        with pytest.raises(fir.FirError):
            g.rows_distances(rows[:1], np.zeros((1, 4), np.int32), 10, 5)
