#!/bin/bash
# usage (GPU box): bash tools/prof_dataprep.sh <tag>  -- kernel-trace statistics of the data-preparation bench
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=${1:-r1}
d=gpurun_out/prof_dataprep_$tag
mkdir -p $d
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/bench_dataprep.py > $d/bench.json 2> $d/err.log
echo rc=$?
cp $(find $d -name "*kernel_stats.csv" | head -1) $d/kernel_stats.csv
grep -E "Name|sv::" $d/kernel_stats.csv | cut -c1-170
cat $d/bench.json
