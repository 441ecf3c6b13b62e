#!/bin/bash
# round 6, first A/B: ablations that price a change before it is built (garbage results, timing only)
#   rec8  - 8-byte pair records (16-bit depth key + 15-bit meta word | signed distance) instead of 12 bytes
#   p3na  - pass 3 without its two LDS accumulator atomics
#   exp / expns with SMIL_STOP=1 - the tile kernel cut off after pass 1, with / without the record stores
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r6
{
bash tools/dbg/ab.sh main rec8 p3na   # (variants built from tools/dbg/r6/ablations.patch applied to raster.hip)
for rep in 1 2; do for v in exp expns; do
  echo "== $v stop=1 STICK: $(SMILFIT_LIB=$L/libsmilfit_$v.so SMIL_STOP=1 python tools/raster_probe.py --frames 4096 --quick --reps 5 2>&1 | grep images)"
done; done
} 2>&1 | tee gpurun_out/r6/abl1.txt
