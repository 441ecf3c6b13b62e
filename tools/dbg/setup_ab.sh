#!/bin/bash
# k_raster_setup's duration (rocprofv3 --kernel-trace --stats of the rasteriser probe) for library variants at the cfg2b and cfg3
# launch sizes:   tools/dbg/setup_ab.sh <name>...   ("main" = libsmilfit.so; ablation variants give garbage lists: timing only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib
for v in "$@"; do
  lib=$L/libsmilfit_$v.so; [ "$v" = main ] && lib=$L/libsmilfit.so
  export SMILFIT_LIB=$lib
  for cfg in "stick --frames 4096" "mouse --model SMILy_Mouse_static_joints --frames 256 --views 18 --radius 4.0"; do
    set -- $cfg; name=$1; shift
    out=gpurun_out/setup_ab/${v}_$name; rm -rf $out; mkdir -p $out
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o k -- python3 tools/raster_probe.py "$@" --quick --reps 4 > $out/log.txt 2>&1 < /dev/null
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    python3 -c "import csv; [print('$v $name:', r['Name'][:40], r['Calls'], 'x', round(float(r['AverageNs'])/1e3,1), 'us') for r in csv.DictReader(open('$f')) if 'k_raster' in r['Name']]"
  done
done
