#!/bin/bash
# round 4: timing of library variants on ONE box (tile kernel alone, through tools/raster_probe.py), STICK cfg2b and mouse cfg3 launches:
#   tools/dbg/r4_base.sh <tag> <name>...     (libsmilfit_<name>.so; "main" = libsmilfit.so; a -DDBG_TIMERS build prints its phase timers)
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
tag=$1; shift
{
for rep in 1 2; do for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  echo "== $v STICK 4096"; SMILFIT_LIB=$lib python tools/raster_probe.py --frames 4096 --quick --reps 6
  echo "== $v mouse 256x18"; SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames 256 --views 18 --radius 4.0 --quick --reps 4
done; done
} 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r4/base_$tag.txt
