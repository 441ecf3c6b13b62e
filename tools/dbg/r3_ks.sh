#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -25
