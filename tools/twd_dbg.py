#!/usr/bin/env python3
"""Debug: fused proposed TWD against the oracle on a small case."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
import golden_cases as gc
from oracle_lib import load_oracle
fir = ge.load_package()
orc = load_oracle()
n, ncls = int(sys.argv[1]) if len(sys.argv) > 1 else 70, 5
rows, cls, q, _ = gc.twd_case(seed=31 + n % 11, n=n, d=256, n_classes=ncls)
q = np.concatenate([q[:5], q[5:8] * np.float32(0.05) + rows[[1, n // 2, n - 1]] * np.float32(0.95)])
with fir.Gallery(rows, cls, 0, 0) as g:
    for fc, th in [(32, 0.7), (64, 0.95), (16, 0.3), (128, 0.7), (32, 1.5), (4, 0.9)]:
        exp = [orc.twd_proposed(rows, cls, qi, fc, th, 0) for qi in q]
        print(fc, th, "oracle", exp)
        for mode in ("0", "1", "2"):
            os.environ["FIR_TWD_FUSED"] = mode
            c, u, k = g.twd_proposed(q, fc, th)
            print("   mode", mode, list(zip(c.tolist(), u.tolist(), k.tolist())))
        os.environ["FIR_TWD_FUSED"] = "1"
        one = [tuple(int(x[0]) for x in g.twd_proposed(q[i:i + 1], fc, th)) for i in range(len(q))]
        print("   one-by-one", one)
