#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag>
# 1) kernel-trace statistics of the default bench command, 2) + 3) HBM traffic counters in passes of their own.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/profile_$tag
mkdir -p $out/trace $out/fetch $out/write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-secondary > $out/bench_under_trace.json 2> $out/trace_err.log || { echo trace failed; exit 1; }
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
timeout -s KILL 180 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $out/fetch_out.json 2> $out/fetch_err.log || { echo fetch pass failed; exit 1; }
cp $(find $out/fetch -name "*counter_collection.csv" | head -1) $out/fetch.csv
timeout -s KILL 180 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-graph > $out/write_out.json 2> $out/write_err.log || { echo write pass failed; exit 1; }
cp $(find $out/write -name "*counter_collection.csv" | head -1) $out/write.csv
python3 - <<PY
import csv, collections, json
def agg(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if 'sv::' in r['Kernel_Name']:
            d[(r['Kernel_Name'], r['Grid_Size'] if 'Grid_Size' in r else r.get('Grid_Size_X', ''))].append(float(r['Counter_Value']))
    return {f"{k[0]} | grid {k[1]}": sum(v) / len(v) for k, v in d.items()}
res = {"FETCH_SIZE_raw_avg_per_launch": agg("$out/fetch.csv"), "WRITE_SIZE_raw_avg_per_launch": agg("$out/write.csv")}
json.dump(res, open("$out/traffic_raw.json", "w"), indent=1)
# HBM bytes per launch by kernel name (all launches of a name averaged): FETCH_SIZE is in KB and, on gfx950, reports half
# of the bytes of wide coalesced reads (x2); WRITE_SIZE is in KB and exact for wide stores.
import re
def by_name(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if 'sv::' in r['Kernel_Name']:
            n = re.sub(r'^void ', '', r['Kernel_Name']).split('(')[0].replace('sv::', '')
            d[n].append(float(r['Counter_Value']))
    return {k: sum(v) / len(v) for k, v in d.items()}
fe, wr = by_name("$out/fetch.csv"), by_name("$out/write.csv")
lat = {"unit": "bytes", "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of bench.py --no-graph; "
       "bytes = 2 * FETCH_SIZE[KB] * 1024 + WRITE_SIZE[KB] * 1024 (gfx950: FETCH_SIZE counts half of wide reads)",
       "hbm_bytes_per_launch": {k: round(2 * fe.get(k, 0) * 1024 + wr.get(k, 0) * 1024) for k in set(fe) | set(wr)}}
json.dump(lat, open("$out/traffic_latest.json", "w"), indent=1)
for k, v in res.items():
    print(k)
    for n, x in sorted(v.items(), key=lambda t: -t[1])[:14]:
        print(f"  {x:12.1f}  {n[:110]}")
PY
rm -rf $out/trace $out/fetch $out/write
head -14 $out/kernel_stats.csv | cut -c1-160
