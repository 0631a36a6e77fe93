#!/bin/bash
# HBM traffic of the pricing kernel: two separate --pmc passes (never combined with other trace
# domains), counters collected for that kernel only so the rest of the run keeps its normal speed.
# Same command as the default bench line; the summary averages the LAST 500 dispatches, i.e. the
# launches bench.py times with HIP events for roofline.achieved.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 280 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_price_sell" -d gpurun_out/pmc_$c -o runc -- python bench.py --cpu-iterations 0 > gpurun_out/pmc_$c.log 2>&1
done
ls gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE
