#!/bin/bash
# rasteriser call (setup + tile kernel) at 1 ... 32 images: tools/dbg/small_launches.sh <name>...   ("main" = libsmilfit.so)
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  for n in 1 2 8 32; do
    echo "$v STICK $n: $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames $n --quick --reps 200 2>&1 | grep images)"
  done
  echo "$v mouse 1x18: $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 1 --views 18 --radius 4.0 --quick --reps 100 2>&1 | grep images)"
done
