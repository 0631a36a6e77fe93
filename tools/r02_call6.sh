#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c6_build.log 2>&1
timeout -k 5 900 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider -k "communicator or two_level or smoke or afiro or random_lp or full_size or reinversion or factor or singular or reload or clone or dual_row_pivot or shared_context" > gpurun_out/c6_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c6_pytest.log
timeout -k 5 150 python tools/solve_profile.py --workload sparse --budget 120 --chunk 2000 > gpurun_out/c6_solve_sparse.log 2>&1
tail -4 gpurun_out/c6_pytest.log; tail -3 gpurun_out/c6_solve_sparse.log | cut -c1-250
