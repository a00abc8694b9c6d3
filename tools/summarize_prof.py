#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_*) into the small summaries kept under profiles/.
usage: summarize_prof.py <kernel_stats.csv|counter_collection.csv> <out.txt|out.json>"""
import collections
import csv
import json
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    rows = list(csv.DictReader(open(src)))
    if "Counter_Name" in rows[0]:
        agg = collections.defaultdict(list)
        meta = {}
        for r in rows:
            k = (r["Kernel_Name"], r["Counter_Name"])
            agg[k].append(float(r["Counter_Value"]))
            meta[r["Kernel_Name"]] = {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "lds": r["LDS_Block_Size"],
                                      "grid": r["Grid_Size"], "wg": r["Workgroup_Size"]}
        out = []
        for (kn, cn), v in sorted(agg.items()):
            if not kn.startswith(("void fir::", "fir::", "(anonymous namespace)::k_", "void (anonymous namespace)::k_")):
                continue
            e = {"kernel": kn, "counter": cn, "dispatches": len(v), "avg": sum(v) / len(v), "min": min(v), "max": max(v), **meta[kn]}
            if cn == "FETCH_SIZE" and "k_scan" in kn:
                # rocprofv3 unit: KiB. gfx950: FETCH_SIZE counts 64 B per 128-B request of a 16 B/lane coalesced stream
                # (MI355X_MICROARCH.md, HBM): double it before comparing with a byte count. Only the scan
                # kernels have that access shape; the other kernels' widths are uncalibrated and left raw.
                e["bytes_per_launch_corrected"] = e["avg"] * 1024 * 2
            out.append(e)
        json.dump(out, open(dst, "w"), indent=1)
    else:
        with open(dst, "w") as f:
            f.write("rocprofv3 --kernel-trace --stats : kernel_stats.csv (kernel names cut to 110 chars)\n")
            f.write(f"{'kernel':110s} {'calls':>6s} {'avg_ns':>12s} {'min_ns':>10s} {'max_ns':>10s} {'pct':>7s}\n")
            for r in rows[:14]:
                f.write(f"{r['Name'][:110]:110s} {r['Calls']:>6s} {float(r['AverageNs']):12.1f} {r['MinNs']:>10s} {r['MaxNs']:>10s} {r['Percentage']:>7s}\n")


if __name__ == "__main__":
    main()
