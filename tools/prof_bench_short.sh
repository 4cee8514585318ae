#!/bin/bash
# usage: prof_bench_short.sh <tag> : kernel-trace stats of a short bench run, top kernels only (runs on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$1/bench.json 2> gpurun_out/prof_$1/err.log
echo rc=$?
find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_$1/kernel_stats.csv
head -12 gpurun_out/prof_$1/kernel_stats.csv | cut -c1-150
python3 -c "
import json;d=json.load(open('gpurun_out/prof_$1/bench.json'));print('ms_per_step', d['ms_per_step'], 'value', d['value'])"
