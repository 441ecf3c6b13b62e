#!/bin/bash
# round 3: what the pair sweep (kernel cut off after pass 1: SMIL_STOP=2) pays for its parts.  Variants built with -DRASTER_EXPERIMENT:
# x0 reference, xg no LDS gather conflicts, xd no depth arithmetic, xh no digit counts, xs one lane stores.  Results are garbage by design.
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r3
for rep in 1 2; do for v in x0 xg xd xh xs; do for stop in 0 1 2 99; do
  [ $v != x0 ] && [ $stop != 2 ] && continue
  echo "$v stop=$stop STICK: $(SMIL_STOP=$stop SMILFIT_LIB=$L/libsmilfit_$v.so python tools/raster_probe.py --frames 4096 --quick --reps 5 2>&1 | grep images)"
done; done; done 2>&1 | tee gpurun_out/r3/abl.txt
