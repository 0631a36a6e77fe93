#!/bin/bash
# round 2, GPU call 2: new kernels (two-level MFMA re-inversion, by-row pricing, ratio-test / flip /
# CHUZR changes) -- tests first, then the bench line, whole-solve profiles and a kernel trace.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c2_build.log 2>&1
timeout -k 5 900 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider > gpurun_out/c2_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c2_pytest.log
timeout -k 5 400 python bench.py --steps 2000 --warmup 200 --tto-budget 30 > gpurun_out/c2_bench.log 2>&1
timeout -k 5 120 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c2_bench_driver.log 2>&1
timeout -k 5 100 python tools/solve_profile.py --workload sparse --budget 60 --chunk 2000 > gpurun_out/c2_solve_sparse.log 2>&1
timeout -k 5 60 python tools/solve_profile.py --workload dense --budget 40 --chunk 500 > gpurun_out/c2_solve_dense.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c2_prof" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --cpu-iterations 0 --pmc off --tto-budget 0 > "$GRAFT_REPO_ROOT/gpurun_out/c2_prof.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c2_prof -name "*_results.db" | head -1) "round 2 call 2: rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 (headline + two replay legs)" > gpurun_out/c2_kernel_stats.txt 2>&1
rm -rf gpurun_out/c2_prof
tail -5 gpurun_out/c2_pytest.log; tail -c 1500 gpurun_out/c2_bench.log
