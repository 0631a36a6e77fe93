#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 120 python -m pytest tests -m gpu -x -q -k "verified_refresh_dense" > gpurun_out/c25_tests.log 2>&1
grep -E "passed|failed|Error|assert|^E " gpurun_out/c25_tests.log | tail -8
