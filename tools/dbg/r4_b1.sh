#!/bin/bash
# round 4: per-kernel durations of the one-frame (and eight-frame) fit iteration under rocprofv3: tools/dbg/r4_b1.sh [lib variant]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
v=${1:-main}; lib=$PWD/smilify_amd/lib/libsmilfit_$v.so; [ "$v" = main ] && lib=$PWD/smilify_amd/lib/libsmilfit.so
export SMILFIT_LIB=$lib
out=gpurun_out/r4/b1_$v; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o b1 -- python3 tools/dbg/b1_trace.py > $out/log.txt 2>&1 < /dev/null
echo "rc=$?"
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
tot = 0.0
for r in csv.DictReader(open(f)):
    if int(r["Calls"]) >= 25:
        per = float(r["AverageNs"]) / 1e3 * int(r["Calls"]) / 30.0
        tot += per
        print(f"{r['Name'][:70]:70s} {int(r['Calls']):5d} x {float(r['AverageNs'])/1e3:8.1f} us   {per:7.1f} us / iteration")
print(f"kernel time per iteration {tot:.1f} us")
PY
