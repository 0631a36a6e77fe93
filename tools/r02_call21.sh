#!/bin/bash
# round 2, final validation: the whole GPU suite, then the numbers the docs quote
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > gpurun_out/c21_build.log 2>&1
echo "build+smoke rc $?" >> gpurun_out/c21_build.log
timeout -k 5 700 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider > gpurun_out/c21_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c21_pytest.log
timeout -k 5 300 python bench.py --steps 2000 --warmup 200 > gpurun_out/c21_bench.log 2> gpurun_out/c21_bench.err
timeout -k 5 200 python bench.py --steps 20 --warmup 5 > gpurun_out/c21_bench_driver.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c21_prof" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --cpu-iterations 0 --pmc off --tto-budget 0 > "$GRAFT_REPO_ROOT/gpurun_out/c21_prof.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c21_prof -name "*_results.db" | head -1) "round 2 final chain (flip scatter, tail batches): rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 (headline in hipGraph mode + two eager replay legs of the same 2200 pivots)" > gpurun_out/c21_kernel_stats.txt 2>&1
rm -rf gpurun_out/c21_prof
tail -3 gpurun_out/c21_build.log; grep -E "passed|failed|rc " gpurun_out/c21_pytest.log | tail -4; tail -c 1500 gpurun_out/c21_bench_driver.log; echo; head -20 gpurun_out/c21_kernel_stats.txt
