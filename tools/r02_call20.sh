#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/c20_build.log 2>&1
timeout -k 5 400 python -m pytest tests -m gpu -x -q -k "stepping or full_size or dse or plugin or strong or fast_dual or reload or clone or flip" > gpurun_out/c20_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c20_tests.log | tail -6
for i in 1 2; do timeout -k 5 200 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c20_driver_$i.log 2>&1; done
timeout -k 5 200 python bench.py --steps 2000 --warmup 200 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c20_bench.log 2>&1
for f in gpurun_out/c20_driver_1.log gpurun_out/c20_driver_2.log gpurun_out/c20_bench.log; do python -c "
import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['ms_per_step'], d['roofline'].get('replay_identical'), d['config'].get('pivot_window'))"; done
