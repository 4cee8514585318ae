#!/bin/bash
# usage: tools_prof2.sh <tag> <args for kbench>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/kprof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kprof_$tag -- python3 tools/kbench.py "$@" > gpurun_out/kprof_$tag/out.txt 2> gpurun_out/kprof_$tag/err.log
echo rc=$?
find gpurun_out/kprof_$tag -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/kprof_$tag/kernel_stats.csv
head -12 gpurun_out/kprof_$tag/kernel_stats.csv | cut -c1-150
find gpurun_out/kprof_$tag -name "*kernel_trace.csv" | head -1 | xargs -I{} cp {} gpurun_out/kprof_$tag/kernel_trace.csv
