#!/bin/bash
# round 2, GPU call 1: state of the tree on hardware -- all GPU tests (incl. the new full-size parity
# tests), whole-solve profiles (k growth / time to optimal), the unrun round-1 code (sell_lanes), and a
# kernel-trace baseline of the bench line.
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c1_build.log 2>&1
timeout -k 5 700 python -m pytest tests -m gpu -q -rf --timeout 300 -p no:cacheprovider > gpurun_out/c1_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c1_pytest.log
timeout -k 5 150 python tools/solve_profile.py --workload sparse --budget 110 --chunk 1000 > gpurun_out/c1_solve_sparse.log 2>&1
timeout -k 5 90 python tools/solve_profile.py --workload dense --budget 45 --chunk 500 > gpurun_out/c1_solve_dense.log 2>&1
timeout -k 5 90 python tools/solve_profile.py --workload netlib --budget 45 --chunk 1000 > gpurun_out/c1_solve_netlib.log 2>&1
timeout -k 5 240 bash tools/lanes.sh > gpurun_out/c1_lanes.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c1_prof" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --cpu-iterations 0 > "$GRAFT_REPO_ROOT/gpurun_out/c1_prof.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c1_prof -name "*_results.db" 2>/dev/null | head -1) "round 2 call 1 (round-1 kernels + host fixes): rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0" > gpurun_out/c1_kernel_stats.txt 2>&1
tail -5 gpurun_out/c1_pytest.log
