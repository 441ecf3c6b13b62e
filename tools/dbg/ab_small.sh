#!/bin/bash
# A/B at small launch sizes (tail effects): tools/dbg/ab_small.sh <name>...
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for rep in 1 2; do for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  echo "$v STICK512: $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames 512 --quick --reps 10 2>&1 | grep images)"
  echo "$v STICK64: $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames 64 --quick --reps 20 2>&1 | grep images)"
  echo "$v mouse8x18: $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 8 --views 18 --radius 4.0 --quick --reps 10 2>&1 | grep images)"
  echo "$v mouse512: $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 16 --views 18 --S 512 --radius 4.0 --quick --reps 4 2>&1 | grep images)"
done; done
