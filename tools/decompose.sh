#!/bin/bash
# The 16-row top-1 pass with parts compiled out (audit build, FIR_GEMM_DBG_SKIP: bit 0 no epilogue, 1 no gallery stream, 2 no query-fragment
# re-reads, 5 no MFMAs): kernel time per 2 048 queries at 1M x 512 under rocprofv3 --kernel-trace --stats.   usage: tools/decompose.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
for v in 0 1 3 5 7 33 37 35; do
  FIR_AMD_LIB=$R/fast-image-recognition_amd/libfir_amd_audit.so FIR_GEMM_DBG_SKIP=$v bash $R/tools/kstats_any.sh dbg$v tools/topk_one.py 512 1 > /dev/null 2>&1
  echo -n "DBG_SKIP=$v  "; grep "k_gemm_proxy_f16x" $R/gpurun_out/kstats_dbg$v.txt | head -1 | awk '{for(i=1;i<=NF;i++) if ($i ~ /^[0-9]+$/ && $(i+1) ~ /\./) {print "calls", $i, "avg_ns", $(i+1), "min_ns", $(i+2); break}}'
done
