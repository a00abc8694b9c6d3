#!/bin/bash
# A/B of two builds of the library on one box: alternate processes, same tool, same arguments.
# usage: tools/ab_libs.sh <rounds> <lib A> <lib B> -- <tool and its arguments>     (FIR_AMD_LIB selects the build)
rounds=$1; a=$2; b=$3; shift 4
for r in $(seq 1 $rounds); do
  for lib in "$a" "$b"; do
    echo "== round $r  $(basename $lib)  $*"
    FIR_AMD_LIB=$lib python3 "$@" 2>&1 | tail -3
  done
done
