#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c11_build.log 2>&1
for v in 1 0; do
  CLPGPU_DEBUG_STATS=1 timeout -k 5 120 python tools/flipdbg.py flip_scatter=$v pivots=400 > gpurun_out/c11_dbg_v$v.log 2>&1
  CLPGPU_DEBUG_STATS=1 timeout -k 5 120 python tools/flipdbg.py flip_scatter=$v pivots=4000 > gpurun_out/c11_dbg4k_v$v.log 2>&1
done
grep -h "clpgpu dbg\|pivots\|k_" gpurun_out/c11_dbg*.log | cut -c1-600
