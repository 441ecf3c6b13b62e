#!/bin/bash
# A/B on ONE box, alternating: the fit iteration with the fused LBS kernels (smil_lbs_forward_project / smil_lbs_backward_ndc) and with
# the separate projection / skinning kernels (SMILFIT_UNFUSED_LBS=1).  Prints ms per iteration of the default bench workload.
cd "$GRAFT_REPO_ROOT"
for rep in 1 2 3; do
  for u in 0 1; do
    ms=$(SMILFIT_UNFUSED_LBS=$u python bench.py --steps 20 --warmup 3 --cpu-frames 0 --no-parity 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('%.3f ms/iteration, tile kernel %.3f ms' % (d['ms_per_step'], d['roofline']['kernel_ms']))")
    echo "unfused=$u: $ms"
  done
done
