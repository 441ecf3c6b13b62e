#!/bin/bash
# Per-kernel averages of one bench workload under rocprofv3: tools/dbg/kstats_wl.sh <workload> [bench args]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
w=$1; shift
out=gpurun_out/kstats_wl/$w; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 bench.py --workload $w --steps 6 --warmup 2 --cpu-frames 0 --no-parity "$@" > $out/log.txt 2>&1 < /dev/null
echo "== $w rc=$?"
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["AverageNs"]) > 15000 and int(r["Calls"]) >= 6:
        print(f"  {r['Name'][:50]:50s} {int(r['Calls']):3d} x {float(r['AverageNs'])/1e3:9.1f} us")
PY
