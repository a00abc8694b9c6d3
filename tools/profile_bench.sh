#!/bin/bash
# The round's judged evidence in one go (run on the GPU box through gpurun, from the repository root):
#   1. python bench.py (default flags + the given steps/warmup): the JSON line -> gpurun_out/bench_final.json
#   2. rocprofv3 --kernel-trace --stats of the same bench command without its PMC child, CPU leg, extras and whole-batch
#      verification (so that the table holds the two timed loops only) -> gpurun_out/bench_kernel_stats.txt
# usage: bash tools/profile_bench.sh [steps] [warmup]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
S=${1:-20}
W=${2:-5}
mkdir -p "$R/gpurun_out"
cd "$R"
timeout -k 10 900 python3 bench.py --steps "$S" --warmup "$W" > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err
tail -c 300 gpurun_out/bench_final.json; echo
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/prof_bench"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof_bench" -o kt -- python3 "$R/bench.py" --steps "$S" --warmup "$W" --no-pmc --cpu-seconds 0 --no-extras --no-verify > "$R/gpurun_out/prof_bench.log" 2>&1
F=$(find "$R/gpurun_out/prof_bench" -name 'kt_kernel_stats.csv' | head -1)
python3 "$R/tools/summarize_prof.py" "$F" "$R/gpurun_out/bench_kernel_stats.txt"
head -8 "$R/gpurun_out/bench_kernel_stats.txt" | cut -c1-170
