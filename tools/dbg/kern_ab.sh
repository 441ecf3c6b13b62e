#!/bin/bash
# per-kernel durations of the fit iteration (rocprofv3 --kernel-trace --stats over tools/dbg/b1_trace.py) for library variants
# at several batch sizes:   tools/dbg/kern_ab.sh <tag> "<kernel name pattern>" <name>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/dbg
tag=$1; pat=$2; shift 2
{
for fr in 1 8 64 256; do for v in "$@"; do
  export SMILFIT_LIB=$L/libsmilfit_$v.so; [ "$v" = main ] && export SMILFIT_LIB=$L/libsmilfit.so
  export B1_FRAMES=$fr
  out=gpurun_out/dbg/kab_${tag}_${v}_$fr; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/dbg/b1_trace.py > $out/log.txt 2>&1 < /dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 -c "import csv,sys; [print(\"frames $fr $v:\", r[\"Name\"][:48], round(float(r[\"AverageNs\"])/1e3,1)) for r in csv.DictReader(open(\"$f\")) if \"$pat\" in r[\"Name\"]]"
done; done
} 2>&1 | tee gpurun_out/dbg/kern_ab_$tag.txt
