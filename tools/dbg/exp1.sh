#!/bin/bash
# round-2 experiment 1: how much of the tile kernel's time is the record streams' HBM traffic?
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/exp1
L=$PWD/smilify_amd/lib
P="python tools/raster_probe.py --frames 4096 --quick --reps 4"
{
echo "== baseline lib"; $P
for r in 14 12 10 9 8 6; do echo "== exp lib resident $r"; SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_RESIDENT=$r $P; done
for w in 0x3ff 0xfff; do for r in 14 8; do echo "== exp lib wrap $w resident $r"; SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_WRAP=$w SMIL_RESIDENT=$r $P; done; done
echo "== dbg lib (timers + stats), cfg2b"; SMILFIT_LIB=$L/libsmilfit_dbg.so $P
echo "== dbg lib, mouse 32 frames x 18 views @256"; SMILFIT_LIB=$L/libsmilfit_dbg.so python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 32 --views 18 --radius 4.0 --quick --reps 3
} > gpurun_out/exp1/log.txt 2>&1
tail -40 gpurun_out/exp1/log.txt
