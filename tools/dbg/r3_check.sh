#!/bin/bash
# round 3: GPU test suite, probe timing, per-kernel averages
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/tests.txt 2>&1; rc=$?; tail -6 gpurun_out/r3/tests.txt
[ $rc -ne 0 ] && exit $rc
bash tools/dbg/ab.sh main 2>&1 | tee gpurun_out/r3/ab6.txt
bash tools/dbg/kstats.sh main 2>&1 | head -5 | tee gpurun_out/r3/kstats6.txt
