#!/bin/bash
# round 3: rasteriser fuzz against the oracle's (depth, face id) rule: ordinary scenes and scenes with the camera inside the mesh's reach
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 500 python tools/dbg/fuzz_raster.py 0 40 > gpurun_out/r3/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r3/fuzz.txt
timeout -k 10 500 python tools/dbg/fuzz_raster.py 100 140 clip > gpurun_out/r3/fuzz_clip.txt 2>&1; echo "fuzz clip rc=$?"; grep -c ok gpurun_out/r3/fuzz_clip.txt; grep FAIL gpurun_out/r3/fuzz_clip.txt | head; tail -2 gpurun_out/r3/fuzz_clip.txt
