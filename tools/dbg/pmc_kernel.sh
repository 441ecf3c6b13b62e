#!/bin/bash
# SQ instruction / wait counters of every rasteriser kernel of one raster_probe run (one rocprofv3 --pmc pass):
#   tools/dbg/pmc_kernel.sh <tag> <raster_probe args...>     e.g.  tools/dbg/pmc_kernel.sh tie --frames 4096 --tie-rule reference_queue
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmck_$tag; rm -rf "$out"; mkdir -p "$out"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d "$out/sq" -o sq -- python3 tools/raster_probe.py "$@" --quick --reps 2 > "$out/sq.log" 2>&1 < /dev/null
echo "rc=$?"
python3 - "$(find "$out/sq" -name '*counter_collection.csv' | head -1)" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:34]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k in acc:
    if "raster" in k:
        print(k, {c.replace("SQ_", ""): f"{v / calls[(k, c)]:.3g}" for c, v in sorted(acc[k].items())})
PY
