#!/bin/bash
# usage (on the GPU box): bash tools/prof_models.sh <tag> <dtype> <batch>
# kernel-trace statistics of the two GAN steps (tools/bench_models.py) -> gpurun_out/prof_<tag>_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; dt=$2; b=$3
out=gpurun_out/prof_$tag
mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/bench_models.py --dtype $dt --batch $b --steps 20 > $out/bench.json 2> $out/err.log || { echo trace failed; tail -5 $out/err.log; exit 1; }
cp $(find $out/trace -name "*kernel_stats.csv" | head -1) gpurun_out/prof_${tag}_kernel_stats.csv
rm -rf $out/trace
head -40 gpurun_out/prof_${tag}_kernel_stats.csv | cut -c1-200
