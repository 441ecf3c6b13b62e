#!/bin/bash
# round 3: wave-state / cache / atomic counters of the fused tile kernel at the cfg2b launch size (tools/pmc_diag.sh) -> gpurun_out/r3/diag_cfg2b.txt
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
bash tools/pmc_diag.sh cfg2b --frames 4096 > gpurun_out/r3/diag_cfg2b.txt 2>&1
tail -62 gpurun_out/r3/diag_cfg2b.txt
