import csv, sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
rows.sort()
n=int(sys.argv[2])
rows=rows[-n:]
t0=rows[0][0]
qs=sorted({r[2] for r in rows})
for s,e,q,nm in rows:
    name=nm.replace("void ","").replace("sv::","").split("(")[0][:70]
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f} q{qs.index(q)} {name}")
