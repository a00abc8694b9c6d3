#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the runs behind the bench line's fractions (kept under profiles/ per round):
#   bench     the two timed loops of bench.py (headline default dispatch + exact scan)
#   config5   bench.py's config-5 block alone (1M x 1280, every batch size)
#   config3   chi-square and KL top-1 / top-5 at 1M x 512 through the default dispatch (the nomination scans)
#   k3        the float64 classifiers (PNN / kNN scan, kNN through the matrix cores)
# usage: bash tools/profile_configs.sh   (on the GPU box, from the repository root) -> gpurun_out/kstats_*.txt
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$R"
bash tools/kstats_any.sh bench bench.py --steps 10 --warmup 3 --no-pmc --cpu-seconds 0 --no-extras --no-verify | head -6
bash tools/kstats_any.sh config5 tools/cfg5_bench_block.py 1 | head -8
bash tools/kstats_any.sh config3_chi2 tools/chi2_bench.py --metric 1 | head -8
bash tools/kstats_any.sh config3_kl tools/chi2_bench.py --metric 2 | head -8
bash tools/kstats_any.sh k3 tools/knn_bench.py --queries 4096 | head -8
