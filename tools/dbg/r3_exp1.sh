#!/bin/bash
# round 3, experiment 1: occupancy sensitivity (4 vs 5 waves per SIMD with spills) + work counters of the current kernel
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r3
bash tools/dbg/ab.sh main 4ws 5w > gpurun_out/r3/exp1_ab.txt 2>&1
cat gpurun_out/r3/exp1_ab.txt
SMILFIT_LIB=$L/libsmilfit_dbg.so python tools/raster_probe.py --frames 4096 --quick --reps 2 > gpurun_out/r3/exp1_dbg_stick.txt 2>&1
SMILFIT_LIB=$L/libsmilfit_dbg.so python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --quick --reps 2 > gpurun_out/r3/exp1_dbg_mouse.txt 2>&1
tail -8 gpurun_out/r3/exp1_dbg_stick.txt
SMILFIT_LIB=$L/libsmilfit_5w.so timeout -k 10 400 python -m pytest tests/test_gpu_baseline_configs.py -x -q -m gpu > gpurun_out/r3/exp1_5w_tests.txt 2>&1; tail -3 gpurun_out/r3/exp1_5w_tests.txt
