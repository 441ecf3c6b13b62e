#!/bin/bash
# per-kernel times of the split path (probe workload) + routing counters
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
for cfg in "stick --frames 4096" "mouse --model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0"; do
  set -- $cfg; tag=$1; shift
  out=gpurun_out/r3/ks_$tag; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/raster_probe.py "$@" --quick --reps 4 > $out/log.txt 2>&1 < /dev/null
  echo "== $tag rc=$?"; tail -2 $out/log.txt
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "raster" in r["Name"] or "unpack" in r["Name"]:
        print(f"  {r['Name'][:60]:60s} {int(r['Calls']):3d} x {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
