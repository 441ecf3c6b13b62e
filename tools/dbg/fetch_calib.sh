#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration run on the GPU box: tools/dbg/fetch_calib.sh  (results under gpurun_out/calib/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/calib; rm -rf $out; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $out/fetch_calib tools/dbg/fetch_calib.hip || exit 1
cd $out
timeout -k 10 200 ./fetch_calib > bytes.txt 2>&1 < /dev/null; echo "plain rc=$?"; cat bytes.txt
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d f -o f -- ./fetch_calib > f.log 2>&1 < /dev/null; echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d w -o w -- ./fetch_calib > w.log 2>&1 < /dev/null; echo "write rc=$?"
python3 - <<'PY'
import csv, glob
for tag in ("f", "w"):
    for path in glob.glob(f"{tag}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            print(tag, r["Kernel_Name"][:30], r["Counter_Name"], float(r["Counter_Value"]) * 1024 / 2**30, "GiB (counter KB x 1024)")
PY
