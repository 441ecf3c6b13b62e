#!/bin/bash
# second half of a campaign: the queue mode and a longer clip batch (fuzz_campaign.sh runs default / clip / lbs): tools/dbg/fuzz_campaign2.sh <tag> <seed0>
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/dbg
tag=$1; s0=${2:-1000}
timeout -k 10 450 python tools/dbg/fuzz_raster.py $((s0 + 2000)) $((s0 + 2300)) queue > gpurun_out/dbg/fuzz_queue_$tag.txt 2>&1; echo "fuzz queue rc=$? $(grep -c ' ok' gpurun_out/dbg/fuzz_queue_$tag.txt) ok, $(grep -c FAIL gpurun_out/dbg/fuzz_queue_$tag.txt) FAIL"
timeout -k 10 650 python tools/dbg/fuzz_raster.py $((s0 + 6000)) $((s0 + 6120)) clip > gpurun_out/dbg/fuzz_clip2_$tag.txt 2>&1; echo "fuzz clip rc=$? $(grep -c ' ok' gpurun_out/dbg/fuzz_clip2_$tag.txt) ok, $(grep -c FAIL gpurun_out/dbg/fuzz_clip2_$tag.txt) FAIL"
