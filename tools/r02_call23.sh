#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c23_build.log 2>&1
timeout -k 5 100 python tools/solve_profile.py --workload dense --budget 40 --chunk 2000 > gpurun_out/c23_dense.log 2>&1
tail -2 gpurun_out/c23_dense.log | cut -c1-400
