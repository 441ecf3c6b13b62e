#!/bin/bash
# round 4: PC sampling of the tile kernel (rocprofv3 beta feature): tools/dbg/r4_pcs.sh <tag> <method> <unit> <interval> <probe args...>
tag=$1; method=$2; unit=$3; interval=$4; shift 4
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r4/pcs_$tag; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
    --kernel-trace --output-format csv -d $out -o pcs -- python3 tools/raster_probe.py "$@" --quick --reps 3 > $out/log.txt 2>&1 < /dev/null
echo "rc=$?"; tail -5 $out/log.txt; find $out -type f | head; for f in $(find $out -name "*.csv"); do echo "$f: $(wc -l < $f) lines"; head -3 $f | cut -c1-300; done
