#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c24_build.log 2>&1
timeout -k 5 100 python tools/solve_profile.py --workload dense --budget 40 --chunk 2000 --opts log_level=2 > gpurun_out/c24_dense.log 2>&1
grep "clpgpu:" gpurun_out/c24_dense.log | tail -6 | cut -c1-200
grep summary gpurun_out/c24_dense.log | cut -c1-400
timeout -k 5 150 python -m pytest tests -m gpu -x -q -k "dense or verified_refresh" > gpurun_out/c24_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c24_tests.log | tail -4
