#!/bin/bash
# HBM traffic of the fused tile kernel from rocprofv3 PMC counters (separate passes; program directly after "--").
#   tools/pmc_traffic.sh <tag> <probe args...>      e.g.  tools/pmc_traffic.sh cfg2b --frames 4096
# Writes gpurun_out/pmc_<tag>/{fetch,write,sq}/... csv; tools/pmc_summary.py turns them into profiles/r2_traffic.json.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag; mkdir -p $out
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  set -- $pass "$@"; name=$1; shift; ctrs=""; while [ "$1" != "--frames" ] && [ "$1" != "--model" ] && [ -n "$1" ]; do ctrs="$ctrs $1"; shift; done
  timeout -k 10 500 rocprofv3 --pmc $ctrs --output-format csv -d $out/$name -o $name -- python3 tools/raster_probe.py "$@" --quick --reps 2 > $out/$name.log 2>&1
  echo "$name done: $(find $out/$name -name '*counter_collection.csv' | head -1)"
done
