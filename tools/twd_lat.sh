#!/bin/bash
# TWD call latency (proposed: prop / prop_all; conventional: conv = posteriors, conv1 = distance difference, conv2 = ratio): the one-launch form (default for up to 8 queries) against the launch-per-chunk forms (FIR_TWD_FUSED=0),
# early exit (prop) and all eight chunks (prop_all) -> stdout
R=${GRAFT_REPO_ROOT:-$(pwd)}
for shape in "3030 1536" "100000 512" "1000000 512"; do
  for qb in 1 8; do
    for which in prop prop_all conv conv1 conv2; do
      for mode in 1 0; do
        FIR_TWD_FUSED=$mode timeout -k 10 120 python3 "$R/tools/twd_probe.py" $shape $qb $which 2>/dev/null | tail -1
      done
    done
  done
done
