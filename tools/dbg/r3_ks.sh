#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
bash tools/dbg/kstats.sh main 2>&1 | head -4
