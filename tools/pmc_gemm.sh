#!/bin/bash
# Hardware counters of the matrix-core nomination kernel at the headline shape (1M x 512, 32 768-query calls), in separate
# rocprofv3 --pmc passes (never combined with a trace domain), merged into one JSON -> gpurun_out/pmc_gemm_f16x.json
# usage: bash tools/pmc_gemm.sh [dim]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=${1:-512}
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS" \
           "FETCH_SIZE TCC_EA0_RDREQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_TA_BUSY_sum"; do
  i=$((i+1))
  rm -rf "$R/gpurun_out/pmc_gemm_$i"
  timeout -k 10 240 rocprofv3 --pmc $SET --output-format csv -d "$R/gpurun_out/pmc_gemm_$i" -o p -- python3 "$R/tools/mfma_ab.py" --dim "$D" --rounds 2 --check 0 --variants "default:" > "$R/gpurun_out/pmc_gemm_$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$R/gpurun_out/pmc_gemm_$i.log"; continue; }
  F=$(find "$R/gpurun_out/pmc_gemm_$i" -name 'p_counter_collection.csv' | head -1)
  python3 "$R/tools/summarize_prof.py" "$F" "$R/gpurun_out/pmc_gemm_$i.json"
done
python3 - "$R" <<'PY'
import json, sys, glob
R = sys.argv[1]
out = {}
for f in sorted(glob.glob(R + "/gpurun_out/pmc_gemm_[0-9]*.json")):
    for e in json.load(open(f)):
        if "k_gemm_proxy_f16x" not in e["kernel"]:
            continue
        k = out.setdefault(e["kernel"][:80], {"meta": {m: e[m] for m in ("vgpr", "sgpr", "lds", "grid", "wg")}, "counters": {}})
        k["counters"][e["counter"]] = {"avg": e["avg"], "min": e["min"], "max": e["max"], "dispatches": e["dispatches"]}
for k, v in out.items():
    c = {n: x["avg"] for n, x in v["counters"].items()}
    d = {}
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs (cycles); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* in quad-cycles
    if "GRBM_GUI_ACTIVE" in c:
        d["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        d["mfma_pipe_busy_fraction"] = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024) / (c["GRBM_GUI_ACTIVE"] / 8)
    if "SQ_INSTS_MFMA" in c:
        m = c["SQ_INSTS_MFMA"]
        d["mfma_per_launch"] = m
        for n_ in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM"):
            if n_ in c:
                d[n_.lower() + "_per_mfma"] = (c[n_] - (m if n_ == "SQ_INSTS_VALU" else 0)) / m
    if "SQ_WAVE_CYCLES" in c:
        for n_ in ("SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if n_ in c:
                d[n_.lower() + "_over_wave_cycles"] = c[n_] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c:
        d["hbm_bytes_per_launch_fetch_size_x1024_x2"] = c["FETCH_SIZE"] * 1024 * 2
    if "SQ_INST_LEVEL_VMEM" in c and "SQ_INSTS_VMEM" in c:
        d["vmem_instruction_latency_level_over_insts"] = c["SQ_INST_LEVEL_VMEM"] / c["SQ_INSTS_VMEM"]
    if "SQ_INST_LEVEL_LDS" in c and "SQ_INSTS_LDS" in c:
        d["lds_instruction_latency_level_over_insts"] = c["SQ_INST_LEVEL_LDS"] / c["SQ_INSTS_LDS"]
    if "TCP_TCC_READ_REQ_LATENCY_sum" in c and "TCP_TCC_READ_REQ_sum" in c and c["TCP_TCC_READ_REQ_sum"]:
        d["l1_to_l2_read_latency_cycles"] = c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"]
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["sq_wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    v["derived"] = d
json.dump(out, open(R + "/gpurun_out/pmc_gemm_f16x.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1))
PY
