#!/bin/bash
# round 6, second pricing run: (a) pass 1 with / without its record stores (cut-off after pass 1: what the drain of the stores in the one
# in-order vmcnt costs at most), (b) where the replay kernel's wave cycles go
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r6
{
for rep in 1 2; do for v in exp expns; do
  echo "== $v stop=2 STICK: $(SMILFIT_LIB=$L/libsmilfit_$v.so SMIL_STOP=2 python tools/raster_probe.py --frames 4096 --quick --reps 5 2>&1 | grep images)"
done; done
echo "== tie timers STICK 4096"
SMILFIT_LIB=$L/libsmilfit_tt.so python tools/dbg/tie_timers.py --frames 4096 --reps 3 2>&1 | tail -12
echo "== tie timers mouse 64 x 18"
SMILFIT_LIB=$L/libsmilfit_tt.so python tools/dbg/tie_timers.py --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --reps 3 2>&1 | tail -12
} 2>&1 | tee gpurun_out/r6/abl2.txt
