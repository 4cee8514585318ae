#!/bin/bash
# usage (GPU box): bash tools/round_measure.sh <tag>
# Every table the round's documents quote, in one go, into gpurun_out/<tag>_final/: the bench line, the secondary steps
# (both policies, the drivers' batch sizes and batch 1024), the per-layer tables, the BatchNorm table, the bf16 error table.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/${tag}_final
mkdir -p $out
set -e
timeout -k 10 300 python3 bench.py --steps 200 --warmup 20 > $out/bench.json 2> $out/bench.err
: > $out/bench_models.jsonl
for cfg in "f32 0" "bf16 0" "f32 1024" "bf16 1024"; do
  set -- $cfg
  timeout -k 10 300 python3 tools/bench_models.py --steps 40 --dtype $1 --batch $2 2>/dev/null >> $out/bench_models.jsonl
done
timeout -k 10 300 python3 tools/gemmbench.py --batch 1024 > $out/gemmbench_b1024.txt 2>&1
timeout -k 10 300 python3 tools/gemmbench.py --batch 64 > $out/gemmbench_b64.txt 2>&1
timeout -k 10 300 python3 tools/gemmbench.py --batch 1024 --dtype bf16 > $out/gemmbench_bf16_b1024.txt 2>&1
timeout -k 10 300 python3 tools/gemmbench.py --batch 64 --dtype bf16 > $out/gemmbench_bf16_b64.txt 2>&1
timeout -k 10 300 python3 tools/bnbench.py > $out/bnbench_f32.txt 2>&1
timeout -k 10 300 python3 tools/bnbench.py --dtype bf16 > $out/bnbench_bf16.txt 2>&1
timeout -k 10 300 python3 tools/bf16_errors.py > $out/bf16_errors.txt 2>&1
cut -c1-400 $out/bench.json; cat $out/bench_models.jsonl | cut -c1-260
