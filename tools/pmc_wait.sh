#!/bin/bash
# usage (GPU box): bash tools/pmc_wait.sh   - wave-cycle accounting of the headline step's kernels (one --pmc pass)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_wait
mkdir -p $out/p1
timeout -s KILL 60 rocprofv3 -L > $out/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $out/counters.txt | sort -u | tr '\n' ' ' > $out/sq_names.txt
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU --output-format csv -d $out/p1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-graph > $out/p1.out 2> $out/p1.err || { echo pass failed; tail -5 $out/p1.err; exit 1; }
python3 - <<PY
import csv, collections, glob
f = glob.glob("$out/p1/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if 'sv::' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(k, {c: round(sum(x) / len(x)) for c, x in v.items()})
PY
