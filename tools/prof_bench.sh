#!/bin/bash
# usage: tools_prof.sh <tag>   (runs on the GPU box) kernel-trace stats of a short bench run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/prof_$1/bench.json 2> gpurun_out/prof_$1/err.log
echo rc=$?
find gpurun_out/prof_$1 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/prof_$1/kernel_stats.csv
head -30 gpurun_out/prof_$1/kernel_stats.csv | cut -c1-200
