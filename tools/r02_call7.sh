#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c7_build.log 2>&1
timeout -k 5 700 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider -k "not 8192 and not large_nucleus and not dense_5000 and not 1500_pivots" > gpurun_out/c7_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c7_pytest.log
timeout -k 5 200 python bench.py --steps 2000 --warmup 200 --tto-budget 0 --pmc off > gpurun_out/c7_bench.log 2>&1
timeout -k 5 100 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c7_bench_driver.log 2>&1
tail -4 gpurun_out/c7_pytest.log; tail -c 600 gpurun_out/c7_bench_driver.log
