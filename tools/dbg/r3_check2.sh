#!/bin/bash
# round 3: raster test modules, then probe timing and per-kernel averages
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_round2.py tests/test_gpu_baseline_configs.py tests/test_gpu_edge_cases.py -x -q -m gpu > gpurun_out/r3/tests2.txt 2>&1; rc=$?; tail -12 gpurun_out/r3/tests2.txt
[ $rc -ne 0 ] && exit $rc
bash tools/dbg/ab.sh main 2>&1 | tee gpurun_out/r3/ab2.txt
bash tools/dbg/kstats.sh main 2>&1 | tee gpurun_out/r3/kstats2.txt
