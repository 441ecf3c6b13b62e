#!/bin/bash
# HBM traffic + SQ instruction counts of the fused tile kernel from rocprofv3 PMC counters: one pass per counter group,
# counters only, program directly after "--".
#   tools/pmc_traffic.sh <tag> <probe args...>      e.g.  tools/pmc_traffic.sh cfg2b --frames 4096
# Writes gpurun_out/pmc_<tag>/{fetch,write,sq}/... csv; tools/pmc_summary.py turns them into profiles/<round>_traffic.json
# (and refuses when a pass is missing).  A failing pass does not stop the others.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag; rm -rf "$out"; mkdir -p "$out"
names=(fetch write sq)
counters=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY")
for i in 0 1 2; do
  name=${names[$i]}
  read -r -a ctrs <<< "${counters[$i]}"
  timeout -k 10 500 rocprofv3 --pmc "${ctrs[@]}" --output-format csv -d "$out/$name" -o "$name" -- python3 tools/raster_probe.py "$@" --quick --reps 2 > "$out/$name.log" 2>&1 < /dev/null
  echo "$name rc=$?: $(find "$out/$name" -name '*counter_collection.csv' | head -1)"
done
