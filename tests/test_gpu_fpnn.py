"""GPU parity of FPNNClassifier (qt_cpp/classification.cpp:618-791): coefficients, log-scores and classes against the
REAL reference's outputs (tests/golden) and against the oracle on fresh data."""
import os

import numpy as np
import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_outputs.npz"))

# The coefficients are sums of cos/sin of the device's libm (<= 2 ulp) where the reference used glibc's; each term is
# at most (J-j)/(J(J+1)) <= 0.25 in magnitude, so a few 1e-16 absolute per coefficient.
A_ATOL = 5e-15
# outputs are float sums of d fast-log terms; a last-bit difference in `probab` can move one term by one float ulp
OUT_RTOL, OUT_ATOL = 2e-6, 1e-4


def golden_cases():
    x, lab, ncls = gc.cls_case()
    x2, lab2, ncls2 = gc.fpnn_case2()
    return (("fpnn", x, ncls, GOLD["cls/train"], GOLD["cls/train_class"], GOLD["cls/test"], GOLD["cls/avg"], GOLD["cls/std"]),
            ("fpnn2", x2, ncls2, GOLD["fpnn2/train"], GOLD["fpnn2/train_class"], GOLD["fpnn2/test"], GOLD["fpnn2/avg"], GOLD["fpnn2/std"]))


def test_reference_model_and_decisions_reproduced(fir):
    for tag, x, nc, train, tcls, test, avg, sd in golden_cases():
        for sc in gc.FPNN_SCALES:
            m = fir.Fpnn(x[train], tcls, nc, avg, sd, sc)
            assert m.J == int(GOLD[f"{tag}/{sc}/J"])
            a = m.model()
            ga = GOLD[f"{tag}/{sc}/a"]
            assert a.shape == ga.shape and np.max(np.abs(a - ga)) <= A_ATOL, (tag, sc, np.max(np.abs(a - ga)))
            best, _ = m.predict(x[test])
            assert np.array_equal(best, GOLD[f"{tag}/{sc}/bf"]), (tag, sc)
            for ratio in gc.FPNN_RATIOS:
                bs, _ = m.predict_seq(x[test], ratio)
                assert np.array_equal(bs, GOLD[f"{tag}/{sc}/seq_{ratio}"]), (tag, sc, ratio)
            m.close()


@pytest.mark.parametrize("seed,n,d,ncls,per_class", [(71, 900, 256, 30, 20), (72, 400, 33, 5, 64), (73, 3000, 64, 300, 5), (74, 130, 200, 2, 50)])
def test_matches_oracle_on_fresh_data(fir, oracle, seed, n, d, ncls, per_class):
    x, lab, _ = gc.cls_case(seed=seed, n=n, d=d, n_classes=ncls)
    train = np.concatenate([np.nonzero(lab == c)[0][:per_class] for c in range(ncls)])
    test = np.concatenate([np.nonzero(lab == c)[0][per_class:per_class + 3] for c in range(ncls)])[:70]   # > one internal batch of 64
    tcls = lab[train]
    _, _, avg, sd = oracle.train_stats(x[train])
    sd[d // 2] = 0.0                                           # a constant feature: normalize() maps it to 0 (:647)
    for sc in (1.0, 0.33, 4.0):                                # 4.0 drives many values into the +-0.5 clamp
        J, ea = oracle.fpnn_train(x[train], tcls, ncls, avg, sd, sc)
        m = fir.Fpnn(x[train], tcls, ncls, avg, sd, sc)
        assert m.J == J
        a = m.model()
        assert np.max(np.abs(a - ea)) <= A_ATOL
        best, outs = m.predict(x[test])
        exp = [oracle.fpnn_predict(ea, J, ncls, avg, sd, sc, x[r]) for r in test]
        for i, e in enumerate(exp):
            assert np.allclose(outs[i], e[1], rtol=OUT_RTOL, atol=OUT_ATOL), (sc, i)
            gap = np.sort(e[1])[-1] - np.sort(e[1])[-2]
            if gap > 2 * OUT_ATOL:                             # away from a score tie the class is the reference's
                assert best[i] == e[0], (sc, i)
        assert np.mean(best == np.array([e[0] for e in exp])) > 0.98
        for ratio in (0.9, 0.99, 0.5):
            bs, chunks = m.predict_seq(x[test], ratio)
            es = [oracle.fpnn_predict(ea, J, ncls, avg, sd, sc, x[r], True, ratio) for r in test]
            agree = np.mean((bs == np.array([e[0] for e in es])) & (chunks == np.array([e[2] for e in es])))
            assert agree > 0.97, (sc, ratio, agree)            # a pruning decision can sit on a threshold tie
        m.close()


def test_argument_errors(fir):
    x, lab, ncls = gc.cls_case(seed=75, n=60, d=16, n_classes=3)
    order = np.argsort(lab, kind="stable")
    avg, sd = x.mean(0), x.std(0)
    with pytest.raises(fir.FirError):
        fir.Fpnn(x, lab, ncls, avg, sd)                        # classes not grouped
    with pytest.raises(fir.FirError):
        fir.Fpnn(x[order], lab[order], 2, avg, sd)             # label outside [0, C)
    m = fir.Fpnn(x[order], lab[order], ncls, avg, sd)
    best, outs = m.predict(np.empty((0, 16)))
    assert best.size == 0 and outs.shape == (0, ncls)
    m.close()
