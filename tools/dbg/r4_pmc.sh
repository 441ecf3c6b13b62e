#!/bin/bash
# round 4: one rocprofv3 --pmc pass over the rasteriser probe, per-launch means of the tile kernel's counters:
#   tools/dbg/r4_pmc.sh <tag> "<counter> <counter> ..." <probe args...>     (SMILFIT_LIB / SMIL_STOP from the environment)
tag=$1; ctrs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/pmc_$tag; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $out -o p -- python3 tools/raster_probe.py "$@" --quick --reps 2 > $out/log.txt 2>&1 < /dev/null
echo "$tag rc=$?"
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_raster_dense" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:36s} {sum(v)/len(v):.4e}   (n={len(v)})")
PY
