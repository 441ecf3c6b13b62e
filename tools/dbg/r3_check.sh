#!/bin/bash
# round 3: GPU test suite, then the default bench line
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/tests.txt 2>&1; rc=$?; tail -6 gpurun_out/r3/tests.txt
[ $rc -ne 0 ] && exit $rc
python bench.py > gpurun_out/r3/bench.json 2> gpurun_out/r3/bench.err; echo "bench rc=$?"; tail -3 gpurun_out/r3/bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3/bench.json"))
print({k:d[k] for k in ("value","ms_per_step")}, d["roofline"]["kernel_ms"], d["roofline"]["frac"], d.get("parity_check"), d["cpu_baseline"]["value"])
PY
