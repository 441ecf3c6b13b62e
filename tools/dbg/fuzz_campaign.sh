#!/bin/bash
# long fuzz of the final kernels (the CPU oracle on the box's host cores sets the pace): tools/dbg/fuzz_campaign.sh <tag> <seed0>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dbg
tag=$1; s0=${2:-1000}
timeout -k 10 500 python tools/dbg/fuzz_raster.py $s0 $((s0 + 400)) > gpurun_out/dbg/fuzz_$tag.txt 2>&1; echo "fuzz rc=$? $(grep -c ' ok' gpurun_out/dbg/fuzz_$tag.txt) ok, $(grep -c FAIL gpurun_out/dbg/fuzz_$tag.txt) FAIL"
timeout -k 10 400 python tools/dbg/fuzz_raster.py $((s0 + 5000)) $((s0 + 5250)) clip > gpurun_out/dbg/fuzz_clip_$tag.txt 2>&1; echo "fuzz clip rc=$? $(grep -c ' ok' gpurun_out/dbg/fuzz_clip_$tag.txt) ok, $(grep -c FAIL gpurun_out/dbg/fuzz_clip_$tag.txt) FAIL"
timeout -k 10 200 python tools/dbg/fuzz_lbs.py $s0 $((s0 + 250)) > gpurun_out/dbg/fuzz_lbs_$tag.txt 2>&1; echo "fuzz lbs rc=$?"; tail -1 gpurun_out/dbg/fuzz_lbs_$tag.txt
