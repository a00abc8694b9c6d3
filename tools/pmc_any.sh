#!/bin/bash
# Hardware counters of one kernel family under any python tool, in separate rocprofv3 --pmc passes (never combined with a
# trace domain), merged into one JSON -> gpurun_out/pmc_<tag>.json
# usage: bash tools/pmc_any.sh <tag> <kernel-name substring> <tool.py> [tool args...]
set -e -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; KERN=$2; TOOL=$3; shift 3
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" \
           "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf "$R/gpurun_out/pmc_${TAG}_$i"
  timeout -k 10 300 rocprofv3 --pmc $SET --output-format csv -d "$R/gpurun_out/pmc_${TAG}_$i" -o p -- python3 "$R/$TOOL" "$@" > "$R/gpurun_out/pmc_${TAG}_$i.log" 2>&1 || { echo "pass $i failed"; tail -3 "$R/gpurun_out/pmc_${TAG}_$i.log"; continue; }
  F=$(find "$R/gpurun_out/pmc_${TAG}_$i" -name 'p_counter_collection.csv' | head -1)
  python3 "$R/tools/summarize_prof.py" "$F" "$R/gpurun_out/pmc_${TAG}_$i.json"
  rm -rf "$R/gpurun_out/pmc_${TAG}_$i"
done
python3 - "$R" "$TAG" "$KERN" <<'PY'
import json, sys, glob
R, TAG, KERN = sys.argv[1:4]
out = {}
for f in sorted(glob.glob(R + f"/gpurun_out/pmc_{TAG}_[0-9]*.json")):
    for e in json.load(open(f)):
        if KERN not in e["kernel"]:
            continue
        k = out.setdefault(e["kernel"][:100], {"meta": {m: e[m] for m in ("vgpr", "sgpr", "lds", "grid", "wg")}, "counters": {}})
        k["counters"][e["counter"]] = {"avg": e["avg"], "min": e["min"], "max": e["max"], "dispatches": e["dispatches"]}
for k, v in out.items():
    c = {n: x["avg"] for n, x in v["counters"].items()}
    d = {}
    # GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are quad-cycles summed over the waves
    if "GRBM_GUI_ACTIVE" in c:
        d["kernel_cycles"] = c["GRBM_GUI_ACTIVE"] / 8
    if "SQ_WAVE_CYCLES" in c:
        for n_ in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
            if n_ in c:
                d[n_.lower() + "_over_wave_cycles"] = c[n_] / c["SQ_WAVE_CYCLES"]
    if "SQ_ACTIVE_INST_VALU" in c and "GRBM_GUI_ACTIVE" in c:
        # quad-cycles of VALU issue summed over all waves / (1024 SIMDs x kernel cycles / 4)
        d["valu_busy_fraction_of_simd_cycles"] = c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8)
    if "FETCH_SIZE" in c:
        d["hbm_bytes_per_launch_fetch_size_x1024_x2"] = c["FETCH_SIZE"] * 1024 * 2
    if "SQ_INST_LEVEL_VMEM" in c and "SQ_INSTS_VMEM" in c and c["SQ_INSTS_VMEM"]:
        d["vmem_instruction_latency_level_over_insts"] = c["SQ_INST_LEVEL_VMEM"] / c["SQ_INSTS_VMEM"]
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    for n_ in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM"):
        if n_ in c:
            d[n_.lower()] = c[n_]
    v["derived"] = d
json.dump(out, open(R + f"/gpurun_out/pmc_{TAG}.json", "w"), indent=1)
print(json.dumps({k: v["derived"] for k, v in out.items()}, indent=1))
PY
