#!/bin/bash
# usage (GPU box): bash tools/sweep_points.sh <tag>
# ms/step (graph replay, batch 1024) and fraction of the f32 peak at points of the run_model.py:164-219 hyper-parameter grid:
# the default stack, the heaviest point, middle points (one with BatchNorm, one few-channel) -> gpurun_out/<tag>_sweep_points.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
out=gpurun_out/${tag}_sweep_points.txt
echo "# point = initial_filters,conv_layers,multiplier,conv_operations,kernel_size,batch_norm; tools/step_calls.py, batch 1024" > $out
for p in "cae" "sweep:16,4,2,3,5,0" "sweep:16,3,2,2,3,0" "sweep:16,4,2,2,3,1" "sweep:16,3,1,1,5,0" "sweep:8,3,1,2,3,0" "sweep:8,4,2,1,3,0"; do
  echo "== $p" >> $out
  timeout -k 10 200 python3 tools/step_calls.py $p 2>/dev/null | head -8 >> $out || { echo "failed: $p"; exit 1; }
done
grep -E "^==|ms/step" $out
