#!/bin/bash
# phase timers of the tile kernel (-DTILE_TIMERS variants): tools/dbg/r5_tt.sh <variant>...   STICK 4096 images and the mouse 64 x 18
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for v in "$@"; do
echo "== $v"
SMILFIT_LIB=$L/libsmilfit_$v.so python tools/raster_probe.py --frames 4096 --quick --reps 3 2>&1 | grep -E "images|tile timers" | tail -2
SMILFIT_LIB=$L/libsmilfit_$v.so python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --quick --reps 3 2>&1 | grep -E "images|tile timers" | tail -2
done
