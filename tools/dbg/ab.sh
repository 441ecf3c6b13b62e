#!/bin/bash
# A/B timing of library builds on ONE box, alternating: tools/dbg/ab.sh <name>... (libsmilfit_<name>.so; "main" = libsmilfit.so)
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for rep in 1 2; do for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  echo "$v STICK: $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames 4096 --quick --reps 6 2>&1 | grep images)"
  echo "$v mouse: $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --quick --reps 4 2>&1 | grep images)"
done; done
