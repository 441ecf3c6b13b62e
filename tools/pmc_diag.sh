#!/bin/bash
# Stall / cache diagnostics of the fused tile kernel: several rocprofv3 --pmc passes (counters only, program directly after "--").
#   tools/pmc_diag.sh <tag> <probe args...>     e.g. tools/pmc_diag.sh cfg2b --frames 4096
# Prints the per-launch mean of every counter for k_raster_dense.
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/diag_$tag; rm -rf $out; mkdir -p $out
passes=(
 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
 "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_THREAD_CYCLES_VALU"
 "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT"
 "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS_ATOMIC SQ_INSTS_BRANCH"
 "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"
 "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"
 "TCC_REQ_sum TCC_READ_sum TCC_WRITE_sum TCC_ATOMIC_sum"
 "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_WRREQ_64B_sum"
 "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"
 "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"
)
i=0
for ctrs in "${passes[@]}"; do
  timeout -k 10 300 rocprofv3 --pmc $ctrs --output-format csv -d $out/p$i -o p -- python3 tools/raster_probe.py "$@" --quick --reps 2 > $out/p$i.log 2>&1 < /dev/null
  echo "pass $i rc=$? ($ctrs)"
  i=$((i+1))
done
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for path in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "k_raster_dense" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:44s} {sum(v)/len(v):.4e}   (n={len(v)})")
PY
