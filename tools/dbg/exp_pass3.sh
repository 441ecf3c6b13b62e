#!/bin/bash
# pass-3 ablations (results are garbage, timing is not): SMIL_STOP=3 no pass 3 at all, 4 no LDS accumulator atomics,
# 5 no global gradient atomics, 99 everything
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for rep in 1 2; do for s in 5 6 99; do
  echo "STICK 4096 stop $s: $(SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_STOP=$s python tools/raster_probe.py --frames 4096 --quick --reps 4 2>&1 | grep images)"
done; done
