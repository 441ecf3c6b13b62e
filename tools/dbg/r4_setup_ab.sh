#!/bin/bash
# round 4: setup-kernel and tile-kernel durations (rocprofv3 --kernel-trace --stats over tools/raster_probe.py) of library variants:
#   tools/dbg/r4_setup_ab.sh <tag> <name>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
tag=$1; shift
{
for v in "$@"; do
  export SMILFIT_LIB=$L/libsmilfit_$v.so; [ "$v" = main ] && export SMILFIT_LIB=$L/libsmilfit.so
  for cfgn in stick mouse mouse512; do
    case $cfgn in
      stick) args="--frames 4096";;
      mouse) args="--model SMILy_Mouse_static_joints --frames 256 --views 18 --radius 4.0";;
      mouse512) args="--model SMILy_Mouse_static_joints --frames 64 --views 18 --radius 4.0 --S 512";;
    esac
    out=gpurun_out/r4/sab_${tag}_${v}_$cfgn; rm -rf $out; mkdir -p $out
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/raster_probe.py $args --quick --reps 6 > $out/log.txt 2>&1 < /dev/null
    f=$(find $out -name "*kernel_stats.csv" | head -1)
    echo "$v $cfgn: setup $(grep k_raster_setup $f | cut -d, -f4 | cut -d. -f1) ns  tile $(grep 'k_raster_dense<2>' $f | cut -d, -f4 | cut -d. -f1) ns"
  done
done
} 2>&1 | tee gpurun_out/r4/setup_ab_$tag.txt
