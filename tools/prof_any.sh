#!/bin/bash
# usage (GPU box): bash tools/prof_any.sh <tag> <python script and args>  -- kernel-trace statistics of any script
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
mkdir -p gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag/out.txt 2> gpurun_out/prof_$tag/err.log
echo rc=$?
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/prof_$tag/kernel_stats.csv
head -${TOP:-22} gpurun_out/prof_$tag/kernel_stats.csv | cut -c1-190
