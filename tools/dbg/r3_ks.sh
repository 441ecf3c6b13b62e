#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r3
for v in main s1 s2; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  out=gpurun_out/r3/ks5_$v; rm -rf $out; mkdir -p $out
  SMIL_STOP=0 SMILFIT_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/raster_probe.py --frames 4096 --quick --reps 4 > $out/log.txt 2>&1 < /dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep k_raster_setup $f | cut -d, -f1-4)"
done
