#!/bin/bash
# round 4: long validation batch of the final kernels on one box: rasteriser fuzz (ordinary scenes + cameras inside the mesh's reach),
# LBS fuzz (narrow + wide), full-size config 5, 300-iteration stability run.  Progress lines keep the call alive.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4
timeout -k 10 1000 python tools/dbg/fuzz_raster.py 0 120 > gpurun_out/r4/fuzz.txt 2>&1; echo "fuzz rc=$? $(grep -c ' ok' gpurun_out/r4/fuzz.txt) ok"; tail -1 gpurun_out/r4/fuzz.txt
timeout -k 10 800 python tools/dbg/fuzz_raster.py 200 280 clip > gpurun_out/r4/fuzz_clip.txt 2>&1; echo "fuzz clip rc=$? $(grep -c ' ok' gpurun_out/r4/fuzz_clip.txt) ok"; tail -1 gpurun_out/r4/fuzz_clip.txt
timeout -k 10 600 python tools/dbg/fuzz_lbs.py 0 80 > gpurun_out/r4/fuzz_lbs.txt 2>&1; echo "fuzz lbs rc=$?"; tail -1 gpurun_out/r4/fuzz_lbs.txt
timeout -k 10 600 python tools/dbg/fuzz_lbs.py 300 340 wide > gpurun_out/r4/fuzz_lbs_wide.txt 2>&1; echo "fuzz lbs wide rc=$?"; tail -1 gpurun_out/r4/fuzz_lbs_wide.txt
timeout -k 10 900 python bench.py --workload cfg5 --steps 3 --warmup 1 --cpu-frames 0 > gpurun_out/r4/bench_cfg5_full.json 2> gpurun_out/r4/bench_cfg5_full.err; echo "cfg5 rc=$?"; cut -c1-300 gpurun_out/r4/bench_cfg5_full.json
timeout -k 10 600 python tools/long_run.py 300 512 > gpurun_out/r4/long_run.txt 2>&1; echo "long rc=$?"; tail -3 gpurun_out/r4/long_run.txt
