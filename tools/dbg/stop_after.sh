#!/bin/bash
# the tile kernel cut off after a phase (-DRASTER_EXPERIMENT build, SMIL_STOP): tools/dbg/stop_after.sh <variant> [stops...]
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/dbg
v=$1; shift; stops=${@:-0 1 2 3 4 5 99}
{
for s in $stops; do
  echo "== $v stop=$s STICK: $(SMILFIT_LIB=$L/libsmilfit_$v.so SMIL_STOP=$s python tools/raster_probe.py --frames 4096 --quick --reps 5 2>&1 | grep images)"
  echo "== $v stop=$s mouse: $(SMILFIT_LIB=$L/libsmilfit_$v.so SMIL_STOP=$s python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 256 --views 18 --radius 4.0 --quick --reps 3 2>&1 | grep images)"
done
} 2>&1 | tee gpurun_out/dbg/stop_$v.txt
