#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c16_build.log 2>&1
timeout -k 5 100 python tools/solve_profile.py --workload sparse --budget 22 --chunk 2000 --opts log_level=2 > gpurun_out/c16_solve.log 2>&1
grep "clpgpu:" gpurun_out/c16_solve.log | tail -40
