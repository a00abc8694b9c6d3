#!/usr/bin/env python3
"""K3 at scale: the float64 PNN / kNN classifiers over a 1M x 512 training set in HBM (bench.py's k3 block alone).
usage: python tools/k3_bench.py [--queries 64]"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=64)
    a = ap.parse_args()
    fir = ge.load_package()
    out = bench.k3_classifiers(fir, torch.device("cuda", 0), a, qb=a.queries)
    for k, v in out.items():
        print(k, v, flush=True)


if __name__ == "__main__":
    main()
