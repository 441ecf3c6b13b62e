#!/bin/bash
# one-image launches: time of setup + tile kernel by the number of pieces a tile is dealt out in (SMIL_SPLIT = log2) and when
# the kernel stops after a phase (SMIL_STOP: 0 list, 1 + staging, 2 + pair sweep, 3 + blend / select, 99 everything)
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for n in 1; do for sp in 0 1 2 3 4 5; do for s in 0 2 99; do
  echo "frames $n split $sp stop $s: $(SMILFIT_LIB=$L/libsmilfit_exp.so SMIL_SPLIT=$sp SMIL_STOP=$s python tools/raster_probe.py --frames $n --quick --reps 50 2>&1 | grep images)"
done; done; done
