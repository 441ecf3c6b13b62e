#!/bin/bash
# per-frame betas through the LBS forward (the neural caller's shape): kernel durations under rocprofv3, STICK and mouse
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
{
for m in SMILy_STICK SMILy_Mouse_static_joints; do
  rm -rf gpurun_out/r6/shape_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/shape_$m -o shape -- python3 tools/shape_blend_probe.py --model $m --frames 4096 2>&1 | grep -E "lbs_forward|shape blend"
  f=$(find gpurun_out/r6/shape_$m -name "*kernel_stats.csv" | head -1)
  head -8 "$f" | cut -d, -f1-4
done
python tools/latency_probe.py 2>&1 | grep "B="
} 2>&1 | tee gpurun_out/r6/shape_blend.txt
