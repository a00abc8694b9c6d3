#!/bin/bash
# rocprofv3 --kernel-trace --stats of any python tool -> gpurun_out/kstats_<tag>.txt (the summary kept under profiles/)
# usage: bash tools/kstats_any.sh <tag> <tool.py> [args...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; TOOL=$2; shift 2
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/kstats_$TAG"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/kstats_$TAG" -o kt -- python3 "$R/$TOOL" "$@" > "$R/gpurun_out/kstats_$TAG.log" 2>&1
F=$(find "$R/gpurun_out/kstats_$TAG" -name 'kt_kernel_stats.csv' | head -1)
python3 "$R/tools/summarize_prof.py" "$F" "$R/gpurun_out/kstats_$TAG.txt"
rm -rf "$R/gpurun_out/kstats_$TAG"
cut -c1-175 "$R/gpurun_out/kstats_$TAG.txt"
