#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_fitter.py tests/test_gpu_driver.py -x -q -m gpu 2>&1 | tail -5
bash tools/dbg/kstats.sh main 2>&1 | tee gpurun_out/r3/kstats4.txt
