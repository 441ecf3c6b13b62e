#!/bin/bash
# round 4: the rasteriser call of a handful of images (setup + tile kernel) against workgroups per CU and pieces per tile
# (-DRASTER_EXPERIMENT build: SMIL_RESIDENT, SMIL_SPLIT): tools/dbg/r4_small.sh <variant> [frames...]
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
v=$1; shift; frames=${@:-1 2 4 8 16 32 64}
{
for fr in $frames; do
  export SMILFIT_LIB=$L/libsmilfit_$v.so; unset SMIL_SPLIT SMIL_RESIDENT
  echo "frames $fr: $(python tools/raster_probe.py --frames $fr --reps 5 2>&1 | grep -o 'work items [0-9]* ([0-9.]* tiles/image)')"
  for res in 2 4 8 16; do
    line="  resident/CU $res:"
    for sp in auto 1 2 3; do
      export SMIL_RESIDENT=$res; unset SMIL_SPLIT; [ $sp != auto ] && export SMIL_SPLIT=$sp
      line="$line  split $sp $(python tools/raster_probe.py --frames $fr --quick --reps 40 2>&1 | grep -o 'time/launch [0-9.]* ms' | cut -d' ' -f2)"
    done
    echo "$line"
  done
done
} 2>&1 | tee gpurun_out/r4/small_$v.txt
