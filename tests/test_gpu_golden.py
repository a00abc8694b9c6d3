"""GPU parity against the committed fixtures (outputs of the REAL reference code,
tests/golden/make_golden.py): the HIP path must reproduce the reference's own answers."""
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("metric", [gc.L2, gc.CHI2, gc.KL])
@pytest.mark.parametrize("shape", gc.MATCH_SHAPES)
def test_match_path_reproduces_reference_outputs(fir, metric, shape):
    seed, n, d = shape
    rows, q = gc.match_case(seed, n, d, metric)
    ranges = [(0, d)] + ([(0, 64), (64, 256)] if d >= 256 else [(0, 32), (5, 39)])
    with fir.Gallery(rows, None, metric, 0) as g:
        for (s, e) in ranges:
            pre = f"match/{gc.METRIC_NAMES[metric]}/{seed}_{n}_{d}/{s}_{e}/"
            idx, dist = g.search_top1(q, s, e)
            kidx, kdist = g.search_topk(q, 5, s, e)
            if metric == gc.KL:   # libm logf vs device logf: 1e-5 relative, identical winner unless a near-tie
                np.testing.assert_allclose(dist, GOLD[pre + "best_dist"], rtol=1e-5, atol=1e-9)
                np.testing.assert_allclose(kdist, GOLD[pre + "top5_dist"], rtol=1e-5, atol=1e-9)
                gd = GOLD[pre + "top5_dist"]
                clear = (gd[:, 1] - gd[:, 0]) > 2e-5 * np.abs(gd[:, 0])
                assert np.array_equal(idx[clear], GOLD[pre + "best_idx"][clear])
            else:
                assert np.array_equal(idx, GOLD[pre + "best_idx"])
                assert np.array_equal(bits(dist), bits(GOLD[pre + "best_dist"]))
                assert np.array_equal(kidx, GOLD[pre + "top5_idx"])
                assert np.array_equal(bits(kdist), bits(GOLD[pre + "top5_dist"]))


@pytest.mark.parametrize("name", sorted(gc.special_cases().keys()))
def test_special_cases_reproduce_reference_outputs(fir, name):
    rows, q, metric = gc.special_cases()[name]
    with fir.Gallery(rows, None, metric, 0) as g:
        idx, dist = g.search_top1(q)
        kidx, kdist = g.search_topk(q, 5)
    assert np.array_equal(idx, GOLD[f"special/{name}/best_idx"])
    assert np.array_equal(bits(dist), bits(GOLD[f"special/{name}/best_dist"]))
    assert np.array_equal(kidx, GOLD[f"special/{name}/top5_idx"])
    assert np.array_equal(bits(kdist), bits(GOLD[f"special/{name}/top5_dist"]))


def test_bruteforce_classifier_reproduces_reference_classes(fir):
    rows, cls, q, ncls = gc.twd_case()
    with fir.Gallery(rows, cls, gc.L2, 0) as g:
        for maxf in (300, 64, 256):
            idx, _ = g.search_top1(q, 0, maxf)
            assert np.array_equal(g.classes_of(idx), GOLD[f"bfclass/{maxf}/class"])
    rows, q = gc.match_case(17, 500, 1536, gc.L2)
    with fir.Gallery(rows, None, gc.L2, 0) as g:
        assert np.array_equal(g.search_top1(q)[0], GOLD["ann_bf/idx"])
