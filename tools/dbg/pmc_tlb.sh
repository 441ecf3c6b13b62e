#!/bin/bash
# Address-translation counters of the fused tile kernel (per-CU UTCL1: requests, hits, misses, stalls), one rocprofv3 --pmc pass per group:
#   tools/dbg/pmc_tlb.sh <tag> <raster_probe args...>       e.g.  tools/dbg/pmc_tlb.sh cfg2b --frames 4096
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/tlb_$tag; rm -rf "$out"; mkdir -p "$out"
names=(req stall)
counters=("TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum")
for i in 0 1; do
  name=${names[$i]}
  read -r -a ctrs <<< "${counters[$i]}"
  timeout -k 10 400 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$out/$name" -o "$name" -- python3 tools/raster_probe.py "$@" --quick --reps 2 > "$out/$name.log" 2>&1 < /dev/null
  echo "$name rc=$?"
  f=$(find "$out/$name" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
for k in acc:
    if "raster" in k:
        print(k, {c: f"{v / calls[(k, c)]:.4g}" for c, v in acc[k].items()})
PY
done
