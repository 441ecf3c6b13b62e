#!/bin/bash
# round 3, experiment 2: split tile kernels (phase A / phase B / fused fallback): parity first, then timing
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r3
timeout -k 10 500 python -m pytest tests/test_gpu_baseline_configs.py tests/test_gpu_kernels.py -x -q -m gpu > gpurun_out/r3/exp2_tests.txt 2>&1; rc=$?; tail -15 gpurun_out/r3/exp2_tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/dbg/ab.sh main > gpurun_out/r3/exp2_ab.txt 2>&1
cat gpurun_out/r3/exp2_ab.txt
