#!/bin/bash
# ablation: time of the tile kernel when it stops after a phase (SMIL_STOP), cfg2b and a mouse launch
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/exp3
L=$PWD/smilify_amd/lib
{
for s in 0 1 2 3 99; do echo "== STICK 4096 stop $s"; SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_STOP=$s python tools/raster_probe.py --frames 4096 --quick --reps 4; done
for s in 0 1 2 3 99; do echo "== mouse 64x18 stop $s"; SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_STOP=$s python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --quick --reps 3; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/exp3/log.txt
cat gpurun_out/exp3/log.txt
