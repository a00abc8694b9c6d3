#!/usr/bin/env python3
"""Print the kernels of the last call in a rocprofv3 kernel-trace CSV (start offset, duration, name): usage trace_last_call.py kt_kernel_trace.csv [gap_us]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
gap = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# calls are separated by host-side gaps > gap us
groups, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur) > gap * 1000:
        groups.append(cur); cur = [e]
    else:
        cur.append(e)
groups.append(cur)
g = groups[-2] if len(groups) > 1 else groups[-1]
t0 = g[0][0]
print(f"{len(groups)} groups; last complete one: {len(g)} kernels, {(max(x[1] for x in g) - t0) / 1000:.1f} us from first start to last end")
for s, e, nme in g:
    print(f"  +{(s - t0) / 1000:8.1f} us  {(e - s) / 1000:8.1f} us  {nme[:110]}")
