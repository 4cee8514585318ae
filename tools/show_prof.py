import json,sys
d=json.load(open(sys.argv[1]))
for r in d["per_call"][:14]:
    g=r["geom"] or {}
    print(f'{r["call"]:28s} {r["avg_us"]:8.1f} us x{r["calls_per_step"]:.0f}', {k:g[k] for k in ("H","W","Cin","Cout","up_h","half") if k in g})
