#!/bin/bash
# round 4: frames per block of the chain kernels (k_pose_fwd, k_chain_bwd): per-kernel durations of the fit iteration at several batch sizes
#   tools/dbg/r4_chain_ab.sh <tag> <name>...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; L=$PWD/smilify_amd/lib; mkdir -p gpurun_out/r4
tag=$1; shift
{
for fr in 1 8 64 512 4096; do for v in "$@"; do
  export SMILFIT_LIB=$L/libsmilfit_$v.so; [ "$v" = main ] && export SMILFIT_LIB=$L/libsmilfit.so
  export B1_FRAMES=$fr
  out=gpurun_out/r4/chain_${tag}_${v}_$fr; rm -rf $out; mkdir -p $out
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 tools/dbg/b1_trace.py > $out/log.txt 2>&1 < /dev/null
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "frames $fr $v: pose_fwd $(grep k_pose_fwd $f | cut -d, -f4 | cut -d. -f1) ns  chain_bwd $(grep k_chain_bwd $f | cut -d, -f4 | cut -d. -f1) ns"
done; done
} 2>&1 | tee gpurun_out/r4/chain_ab_$tag.txt
