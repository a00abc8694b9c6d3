#!/usr/bin/env python3
"""Runs the REAL reference harness (oracle/_ref/harness_reference: testRecognition of ImageTesting.cpp with the reference's
own classes, built by oracle/build_ref.sh) on the synthetic Caltech-like file of golden_cases.write_harness_features and
stores the result lines as tests/golden/harness_ImageTesting.txt; likewise testANN (harness_ann_reference -> harness_ann.txt) and testClassification1 (harness_cls_reference ->
harness_classification.txt, on a harder file). Only meaningful in the build container."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402


def main():
    exe = os.path.join(ROOT, "oracle", "_ref", "harness_reference")
    with tempfile.TemporaryDirectory() as td:
        gc.write_harness_features(os.path.join(td, gc.HARNESS_FEATURES_FILE))
        out = subprocess.run([exe], cwd=td, capture_output=True, text=True, timeout=1200, check=True).stdout
    lines = gc.harness_result_lines(out)
    path = os.path.join(HERE, "harness_ImageTesting.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {path}: {len(lines)} lines")
    print("\n".join(lines))
    # testANN (ann.cpp:24-81): BruteForce + DirectedEnumeration over the imageCountToCheck ratios
    exe = os.path.join(ROOT, "oracle", "_ref", "harness_ann_reference")
    with tempfile.TemporaryDirectory() as td:
        gc.write_harness_features(os.path.join(td, gc.HARNESS_FEATURES_FILE))
        out = subprocess.run([exe], cwd=td, capture_output=True, text=True, timeout=1200, check=True).stdout
    lines = gc.ann_harness_result_lines(out)
    path = os.path.join(HERE, "harness_ann.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {path}: {len(lines)} lines")
    # testClassification1 (classification.cpp:991-1089): nine classifiers, six training fractions, two splits each
    exe = os.path.join(ROOT, "oracle", "_ref", "harness_cls_reference")
    with tempfile.TemporaryDirectory() as td:
        gc.write_harness_features(os.path.join(td, gc.HARNESS_FEATURES_FILE), gc.CLS_HARNESS_SIGNAL)
        out = subprocess.run([exe], cwd=td, capture_output=True, text=True, timeout=1200, check=True).stdout
    lines = gc.cls_harness_result_lines(out)
    path = os.path.join(HERE, "harness_classification.txt")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {path}: {len(lines)} lines")


if __name__ == "__main__":
    main()
