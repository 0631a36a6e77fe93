#!/bin/bash
# HBM traffic of the pricing kernel: two separate --pmc passes (never combined with other trace domains)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace -d gpurun_out/pmc_$c -o runc -- python bench.py --steps 300 --warmup 100 --cpu-iterations 0 > gpurun_out/pmc_$c.log 2>&1
done
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
