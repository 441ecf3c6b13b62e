#!/bin/bash
# Per-kernel averages of the default bench iteration for library variants on ONE box: tools/dbg/kstats.sh <name>...
# (libsmilfit_<name>.so; "main" = libsmilfit.so).  Prints every kernel above 20 us except the tile kernel's line count.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  out=gpurun_out/kstats/$v; rm -rf $out; mkdir -p $out
  export SMILFIT_LIB=$lib
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 bench.py --steps 8 --warmup 2 --cpu-frames 0 --no-others > $out/log.txt 2>&1 < /dev/null
  echo "== $v rc=$?"
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 20000 and int(r["Calls"]) >= 8:
        print(f"  {r['Name'][:44]:44s} {int(r['Calls']):3d} x {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
