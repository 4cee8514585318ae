#!/bin/bash
# usage (here, after tools/round_measure.sh, tools/profile_round.sh and tools/prof_models.sh ran on the GPU box):
#   bash tools/collect_profiles.sh <tag>
# copies the merged results from gpurun_out/ into profiles/ under the names profiles/README.md lists
set -e
t=$1
f=gpurun_out/${t}_final; p=gpurun_out/profile_$t
cp $p/kernel_stats.csv profiles/${t}_kernel_stats.csv
cp $p/bench_under_trace.json profiles/${t}_bench_under_trace.json
cp $p/traffic_raw.json profiles/${t}_traffic_raw.json
cp $p/traffic_latest.json profiles/traffic_latest.json
cp $f/bench.json profiles/${t}_bench.json
cp $f/bench_models.jsonl profiles/${t}_bench_models.jsonl
cp gpurun_out/prof_${t}_f32_kernel_stats.csv profiles/${t}_models_kernel_stats_f32.csv
cp gpurun_out/prof_${t}_bf16_b1024_kernel_stats.csv profiles/${t}_models_kernel_stats_bf16_b1024.csv
for n in gemmbench_b1024 gemmbench_b64 gemmbench_bf16_b1024 gemmbench_bf16_b64 bnbench_f32 bnbench_bf16 bf16_errors; do
  grep -v "amdgpu.ids" $f/$n.txt > profiles/${t}_$n.txt
done
ls -la profiles/ | grep " ${t}_"
