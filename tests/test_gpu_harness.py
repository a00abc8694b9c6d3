"""Drop-in check with the reference's OWN harness: testRecognition / testRecognitionMethod (qt_cpp/ImageTesting.cpp:439-548,
minus the three OpenCV classifiers) compiled unmodified against this repository's host shim (oracle/_ref/harness_dropin,
built by oracle/build_ref.sh -- the reference's db.h stays, "db_features.h" resolves to host/compat, the file-static
counter of :33 becomes the shim's, the classes of :35-288 become one #include) must print the same error rates, recalls
and unreliable ratios as the same harness with the reference's own classes did on the CPU
(tests/golden/harness_ImageTesting.txt, made by tests/golden/make_harness_golden.py)."""
import os
import subprocess

import pytest

import golden_cases as gc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "harness_dropin")
GOLD = os.path.join(ROOT, "tests", "golden", "harness_ImageTesting.txt")


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/harness_dropin not built (needs /root/reference at build time)")
def test_reference_harness_prints_the_reference_results(tmp_path):
    gc.write_harness_features(str(tmp_path / gc.HARNESS_FEATURES_FILE))
    out = subprocess.run([EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = gc.harness_result_lines(out.stdout)
    want = open(GOLD).read().splitlines()
    assert got == want, "\n".join(f"{g!r} != {w!r}" for g, w in zip(got, want) if g != w)
    assert len(want) == 32 and any("unrel=86.747%" in line for line in want)      # eight classifiers, two splits each


ANN_EXE = os.path.join(ROOT, "oracle", "_ref", "harness_ann_dropin")


@pytest.mark.skipif(not os.path.exists(ANN_EXE), reason="oracle/_ref/harness_ann_dropin not built (needs /root/reference at build time)")
def test_reference_ann_harness_prints_the_reference_results(tmp_path):
    """testANN (qt_cpp/ann.cpp:24-81 minus the FLANN method) against compat/ann.h: BruteForce and DirectedEnumeration (greedy
    pivots from the same std::rand stream, false-accept threshold, error rate and checked percentage at 19 ratios)."""
    gc.write_harness_features(str(tmp_path / gc.HARNESS_FEATURES_FILE))
    out = subprocess.run([ANN_EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = gc.ann_harness_result_lines(out.stdout)
    want = open(os.path.join(ROOT, "tests", "golden", "harness_ann.txt")).read().splitlines()
    assert got == want, "\n".join(f"{g!r} != {w!r}" for g, w in zip(got, want) if g != w) + f" ({len(got)} vs {len(want)} lines)"
    assert sum(line.startswith("dem error=") for line in want) >= 19


CLS_EXE = os.path.join(ROOT, "oracle", "_ref", "harness_cls_dropin")


@pytest.mark.skipif(not os.path.exists(CLS_EXE), reason="oracle/_ref/harness_cls_dropin not built (needs /root/reference at build time)")
def test_reference_classification_harness_prints_the_reference_results(tmp_path):
    """testClassification1 (qt_cpp/classification.cpp:991-1089 minus the OpenCV classifiers and the OpenCV PCA step) against
    fir_classification.h + compat/classification_globals.h: kNN-1/3, PNN, PNN with clustering, FPNN x2, sequential PNN,
    sequential FPNN x2; training fractions 5..30 per class, two std::rand-driven splits each; error, sigma, recall."""
    gc.write_harness_features(str(tmp_path / gc.HARNESS_FEATURES_FILE), gc.CLS_HARNESS_SIGNAL)
    out = subprocess.run([CLS_EXE], cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    got = gc.cls_harness_result_lines(out.stdout)
    want = open(os.path.join(ROOT, "tests", "golden", "harness_classification.txt")).read().splitlines()
    assert got == want, "\n".join(f"{g!r} != {w!r}" for g, w in zip(got, want) if g != w) + f" ({len(got)} vs {len(want)} lines)"
    assert sum(line.startswith("fraction=") for line in want) == 54                  # 9 classifiers x 6 fractions
