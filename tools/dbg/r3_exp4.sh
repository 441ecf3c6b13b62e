#!/bin/bash
# occupancy sensitivity of the phase kernels: per-kernel times for library variants (STICK 4096 images)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r3
for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  out=gpurun_out/r3/ks4_$v; rm -rf $out; mkdir -p $out
  SMILFIT_LIB=$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/raster_probe.py --frames 4096 --quick --reps 4 > $out/log.txt 2>&1 < /dev/null
  echo "== $v rc=$? $(grep images $out/log.txt)"
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if ("raster" in r["Name"] or "unpack" in r["Name"]) and int(r["Calls"]) > 1:
        print(f"  {r['Name'][:60]:60s} {int(r['Calls']):3d} x {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
