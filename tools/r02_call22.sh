#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/c22_build.log 2>&1
timeout -k 5 200 python -m pytest tests -m gpu -x -q -k "flip or config4_full_size_first or stepping or degenerate" > gpurun_out/c22_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c22_tests.log | tail -6
timeout -k 5 100 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c22_driver.log 2>&1
python -c "
import json; d=json.loads([l for l in open('gpurun_out/c22_driver.log') if l.startswith('{')][-1]); print(round(d['value'],1), d['roofline'].get('replay_identical'))"
