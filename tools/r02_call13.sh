#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c13_build.log 2>&1
timeout -k 5 400 python -m pytest tests -m gpu -x -q -k "strong or fast_dual or degenerate or warm" > gpurun_out/c13_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c13_tests.log | tail -15
