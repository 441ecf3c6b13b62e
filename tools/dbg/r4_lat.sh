#!/bin/bash
# round 4: small-batch latency of library variants on one box: tools/dbg/r4_lat.sh <tag> <name>...   ("main" = libsmilfit.so)
cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
tag=$1; shift
{
for rep in 1 2; do for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  echo "== $v"; SMILFIT_LIB=$lib python tools/latency_probe.py 2>&1 | grep "B="
done; done
} 2>&1 | tee gpurun_out/r4/lat_$tag.txt
