"""Randomised differential test of the scan entry points against the oracle: gallery sizes around the tile, wave and
kernel-selection boundaries, feature counts that are not multiples of 4, odd feature ranges, ragged query batches, all
three metrics, forced queries-per-pass settings. Fixed seed: the cases are the same on every run."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

L2, CHI2, KL = 0, 1, 2


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def kl_bound(qv, rows, start, end):
    """How far two faithful evaluations of the KL arm (db_features.cpp:33-36) may differ when their logf differ by a
    few ulp: the terms l*log(2l/s) and r*log(2r/s) have opposite signs and nearly cancel for similar vectors, so the
    bound scales with the sum of their MAGNITUDES, not with the distance itself. Per row."""
    l = qv[start:end].astype(np.float64)[None, :]
    r = np.atleast_2d(rows)[:, start:end].astype(np.float64)
    s = l + r
    with np.errstate(divide="ignore", invalid="ignore"):
        tl = np.where((s > 0) & (l > 0), np.abs(l * np.log(2 * l / s)), 0.0)
        tr = np.where((s > 0) & (r > 0), np.abs(r * np.log(2 * r / s)), 0.0)
    return 4 * 6e-8 * (tl + tr).sum(axis=1) / (end - start)


def cases():
    rng = np.random.default_rng(20261004)
    out = []
    sizes = [1, 2, 63, 64, 65, 127, 129, 500, 1023, 1025, 3030, 4097, 9000, 20000, 70001]
    dims = [1, 3, 4, 5, 31, 32, 33, 64, 100, 256, 257, 512, 1030]
    for i in range(60):
        n = int(rng.choice(sizes))
        d = int(rng.choice(dims))
        if n * d > 12_000_000:
            d = 64
        metric = int(rng.choice([L2, L2, CHI2, KL]))
        qb = int(rng.choice([1, 2, 3, 5, 8, 9, 16, 17, 33]))
        if rng.random() < 0.5:
            start, end = 0, d
        else:
            start = int(rng.integers(0, d))
            end = int(rng.integers(start + 1, d + 1))
        qpp = int(rng.choice([-1, -1, 1, 2, 4, 8, 16]))
        out.append((i, n, d, metric, qb, start, end, qpp))
    # very long feature vectors: the 8- and 16-query tiles no longer fit LDS (scalar-operand form of the hand-scheduled kernel)
    out.append((60, 300, 5000, L2, 8, 0, 5000, -1))
    out.append((61, 700, 4800, L2, 16, 0, 4800, 16))
    out.append((62, 200, 2300, L2, 9, 4, 2296, 8))        # 8-query tile between 64 and 144 KiB: the opt-in LDS size
    return out


@pytest.mark.parametrize("case", cases(), ids=lambda c: f"{c[0]}-n{c[1]}-d{c[2]}-m{c[3]}-q{c[4]}-r{c[5]}_{c[6]}-p{c[7]}")
def test_scans_match_oracle(fir, oracle, case):
    i, n, d, metric, qb, start, end, qpp = case
    rows = synth.make_gallery(1000 + i, n, d, metric)
    q, _ = synth.make_queries(1000 + i, rows, qb, metric)
    if n > 70 and qb > 2:
        rows[n - 1] = rows[n // 2]             # a duplicate pair: ties resolve to the lower row
        q[1] = rows[n // 2]
    k = min(5, 8)
    with fir.Gallery(rows, None, metric, 0) as g:
        g.set_tuning(qpp, 0)
        idx, dist = g.search_top1(q, start, end)
        tidx, tdist = g.search_topk(q, k, start, end)
        sub = min(qb, 3)
        allq = g.range_distances(q[:sub], start, end)
        pick = np.stack([np.arange(min(n, 7)), np.arange(min(n, 7))[::-1]]).astype(np.int32)[:, : min(n, 7)]
        rd = g.rows_distances(q[:2] if qb >= 2 else np.repeat(q[:1], 2, 0), pick, start, end)
    for j in range(qb):
        ei, ed = oracle.recognize_bf(rows, q[j], start, end, metric)
        ki, kd = oracle.topk(rows, q[j], start, end, k, metric)
        if metric == KL:                       # the device's logf: distances within 1e-5 (+ the cancellation bound), rows equal away from near-ties
            kb = kl_bound(q[j], rows[np.maximum(ki, 0)], start, end)
            assert abs(float(dist[j]) - float(ed)) <= 1e-5 * abs(float(ed)) + kb[0] + 1e-12, (j, dist[j], ed)
            assert np.all(np.abs(tdist[j].astype(np.float64) - kd) <= 1e-5 * np.abs(kd) + kb + 1e-12), (j, tdist[j], kd)
            if n < 2 or kd[1] - kd[0] > 2e-5 * abs(kd[0]) + 2 * kb[:2].max():
                assert idx[j] == ei and tidx[j, 0] == ki[0]
        else:
            assert idx[j] == ei and bits(dist[j]) == bits(ed), (j, idx[j], ei)
            assert np.array_equal(tidx[j], ki) and np.array_equal(bits(tdist[j]), bits(kd)), j
    for j in range(sub):
        exp = oracle.all_distances(rows, q[j], start, end, metric)
        if metric == KL:
            assert np.all(np.abs(allq[j].astype(np.float64) - exp) <= 1e-5 * np.abs(exp) + kl_bound(q[j], rows, start, end) + 1e-12)
        else:
            assert np.array_equal(bits(allq[j]), bits(exp)), j
    qq = q[:2] if qb >= 2 else np.repeat(q[:1], 2, 0)
    for j in range(2):
        for c, r in enumerate(pick[j]):
            exp = oracle.feature_distance(qq[j], rows[r], start, end, metric)
            if metric == KL:
                assert abs(float(rd[j, c]) - float(exp)) <= 1e-5 * abs(float(exp)) + kl_bound(qq[j], rows[r], start, end)[0] + 1e-12
            else:
                assert bits(rd[j, c]) == bits(exp)
