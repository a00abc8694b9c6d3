#!/usr/bin/env python3
"""bench.py's config-5 block alone, several times in one process: fallback queries per call at every batch.
FIR_GEMM_DEBUG_COUNTS=1 adds the appended rows / bound of every call on stderr. usage: cfg5_bench_block.py [reps=3] [dim=1280]"""
import os, sys, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
import bench
fir = ge.load_package()
dev = torch.device("cuda", 0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 1280
args = types.SimpleNamespace(rows=1_000_000, config5_dim=dim, extra_batch=32768)
ws = torch.cuda.Stream()
for r in range(reps):
    print("=== rep", r, file=sys.stderr, flush=True)
    out = bench.config5(args, fir, dev, ws)
    for b, v in out["batches"].items():
        print(f"rep {r} qb {b:>6s} scan {v['exact_scan_queries_per_s']:.0f} mfma {v['matrix_core_queries_per_s']:.0f} default {v['default_dispatch_queries_per_s']:.0f} "
              f"fallbacks/call {v['fallback_queries_per_call']:.2f} same {v['identical_keys']}", flush=True)
