#!/bin/bash
# round 4: rasteriser call (setup + tile kernel) of 1 ... 512 STICK images and 1 ... 8 mouse frames x 18 views, library variants side by side:
#   tools/dbg/r4_small_ab.sh <tag> <name>...
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
tag=$1; shift
{
for fr in 1 2 4 8 16 32 64 128 512; do
  line="STICK frames $fr:"
  for v in "$@"; do lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
    line="$line  $v $(SMILFIT_LIB=$lib python tools/raster_probe.py --frames $fr --quick --reps 40 2>&1 | grep -o 'time/launch [0-9.]* ms' | cut -d' ' -f2)"; done
  echo "$line"
done
for fr in 1 2 8; do
  line="mouse frames $fr x 18 views:"
  for v in "$@"; do lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
    line="$line  $v $(SMILFIT_LIB=$lib python tools/raster_probe.py --model SMILy_Mouse_static_joints --frames $fr --views 18 --radius 4.0 --quick --reps 20 2>&1 | grep -o 'time/launch [0-9.]* ms' | cut -d' ' -f2)"; done
  echo "$line"
done
} 2>&1 | tee gpurun_out/r4/small_ab_$tag.txt
