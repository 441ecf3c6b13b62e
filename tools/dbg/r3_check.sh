#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r3/tests.txt 2>&1; rc=$?; tail -6 gpurun_out/r3/tests.txt
exit $rc
