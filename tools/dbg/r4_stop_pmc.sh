#!/bin/bash
# round 4: wave-state counters of the tile kernel cut off after each phase (-DRASTER_EXPERIMENT build, SMIL_STOP): which phase the
# waiting belongs to.  tools/dbg/r4_stop_pmc.sh <variant> <probe args...>
v=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export SMILFIT_LIB=$PWD/smilify_amd/lib/libsmilfit_$v.so
out=gpurun_out/r4/stop_pmc_$v; rm -rf $out; mkdir -p $out
for s in 0 1 2 3 99; do
  export SMIL_STOP=$s
  timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $out/s$s -o p -- python3 tools/raster_probe.py "$@" --quick --reps 2 > $out/s$s.log 2>&1 < /dev/null
  echo "stop $s rc=$?"
done
python3 - $out <<'PY'
import csv, glob, sys, collections
for s in (0, 1, 2, 3, 99):
    acc = collections.defaultdict(list)
    for path in glob.glob(sys.argv[1] + f"/s{s}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "k_raster_dense" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"stop {s:2d}: " + "  ".join(f"{k[3:]} {sum(v)/len(v):.3e}" for k, v in sorted(acc.items())))
PY
