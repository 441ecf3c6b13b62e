#!/bin/bash
# round 3: full-size config 5 on one GPU (8192 frames x 18 views @512^2 = 147 456 images per iteration) and a 300-iteration stability run
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python bench.py --workload cfg5 --steps 3 --warmup 1 --cpu-frames 0 > gpurun_out/r3/bench_cfg5_full.json 2> gpurun_out/r3/bench_cfg5_full.err; echo "cfg5 rc=$?"
tail -2 gpurun_out/r3/bench_cfg5_full.err; cut -c1-400 gpurun_out/r3/bench_cfg5_full.json
timeout -k 10 600 python tools/long_run.py 300 512 > gpurun_out/r3/long_run.txt 2>&1; echo "long rc=$?"; tail -5 gpurun_out/r3/long_run.txt
