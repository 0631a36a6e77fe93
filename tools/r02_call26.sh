#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout -k 5 100 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c26_driver.log 2>&1
python -c "
import json; d=json.loads([l for l in open('gpurun_out/c26_driver.log') if l.startswith('{')][-1]); print(round(d['value'],1), d['roofline']['dominant_by_time'])"
tail -3 gpurun_out/c26_driver.log | cut -c1-300
