#!/bin/bash
# usage (GPU box): bash tools/pmc_run.sh <tag> "<counters>" <python script and args>   -- one counter pass, per-kernel averages
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctrs=$2; shift; shift
mkdir -p gpurun_out/pmc_$tag
timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag -- python3 "$@" > gpurun_out/pmc_$tag/out.txt 2> gpurun_out/pmc_$tag/err.log
echo rc=$?
f=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1)
cp $f gpurun_out/pmc_$tag/counters.csv
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/pmc_$tag/counters.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'sv::' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
