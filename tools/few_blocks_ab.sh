#!/bin/bash
# calls of 8 / 16 / 32 / 48 queries against 1M x d: the pass over the live query blocks only (default) against the whole 128-query tile
# (FIR_GEMM_FEW_BLOCKS=0), alternated.   usage: tools/few_blocks_ab.sh <d>
d=$1
for q in 8 16 32 48; do
  for r in 1 2; do
    for fb in 0 1; do
      echo -n "d=$d q=$q FEW_BLOCKS=$fb: "
      FIR_GEMM_FEW_BLOCKS=$fb python3 tools/small_call_trace.py 1000000 $d $q 60 2>&1 | tail -1
    done
  done
done
