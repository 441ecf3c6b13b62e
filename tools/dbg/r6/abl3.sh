#!/bin/bash
# round 6: the hand-scheduled prefetch (exp2 = this tree) against the round-5 kernel (exp), cut off after pass 1 and whole
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r6
{
for rep in 1 2; do for v in exp exp2; do for st in 1 2 99; do
  echo "== $v stop=$st STICK: $(SMILFIT_LIB=$L/libsmilfit_$v.so SMIL_STOP=$st python tools/raster_probe.py --frames 4096 --quick --reps 5 2>&1 | grep images)"
done; done; done
} 2>&1 | tee gpurun_out/r6/abl3.txt
