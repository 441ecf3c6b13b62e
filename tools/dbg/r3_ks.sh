#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_baseline_configs.py tests/test_gpu_edge_cases.py tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -4
bash tools/dbg/ab.sh main 2>&1 | tee gpurun_out/r3/ab5.txt
