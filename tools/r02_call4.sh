#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c4_build.log 2>&1
timeout -k 5 600 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider -k "by_row or singular or full_size or afiro or random_lp or two_level or netlib or degenerate or flip_list or warm" > gpurun_out/c4_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c4_pytest.log
CLPGPU_DEBUG_STATS=1 timeout -k 5 300 python bench.py --steps 2000 --warmup 200 --tto-budget 45 --pmc off > gpurun_out/c4_bench.log 2> gpurun_out/c4_bench.err
timeout -k 5 120 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c4_bench_driver.log 2>&1
tail -4 gpurun_out/c4_pytest.log; grep "clpgpu dbg" gpurun_out/c4_bench.err | head -3; tail -c 2500 gpurun_out/c4_bench.log
