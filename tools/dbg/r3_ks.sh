#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
bash tools/dbg/ab.sh main 2>&1 | tee gpurun_out/r3/ab3.txt
bash tools/dbg/kstats.sh main 2>&1 | head -4 | tee gpurun_out/r3/kstats3.txt
