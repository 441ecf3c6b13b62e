#!/bin/bash
# A/B of library builds under tie_rule = reference_queue (fused rasteriser call at the cfg2b / mouse launch sizes): tools/dbg/ab_tie.sh <name>...
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for rep in 1 2; do for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  echo "$v STICK: $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames 4096 --quick --reps 6 --tie-rule reference_queue 2>&1 | grep images)"
  echo "$v mouse: $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --quick --reps 4 --tie-rule reference_queue 2>&1 | grep images)"
done; done
