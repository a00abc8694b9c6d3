import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
ev=sorted(((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"]) for r in rows))
key=sys.argv[2]
idx=[i for i,e in enumerate(ev) if key in e[2]]
i0=idx[-2]; i1=idx[-1]
t0=ev[i0][0]
for s,e,n in ev[i0:i1]:
    print(f"+{(s-t0)/1000:8.1f} us {(e-s)/1000:8.1f} us  {n[:100]}")
