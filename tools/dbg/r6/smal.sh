#!/bin/bash
# SMAL.__call__ forward + backward with per-frame betas: kernel durations under rocprofv3, STICK and mouse
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r6
for m in SMILy_STICK SMILy_Mouse_static_joints; do
  rm -rf gpurun_out/r6/smal_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r6/smal_$m -o smal -- python3 tools/smal_call_probe.py --model $m --frames 4096 2>&1 | grep "SMAL.__call__"
done
