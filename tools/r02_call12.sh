#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c12_build.log 2>&1
for v in 1 0; do
  CLPGPU_DEBUG_STATS=1 timeout -k 5 120 python tools/flipdbg.py flip_scatter=$v pivots=400 > gpurun_out/c12_dbg_v$v.log 2>&1
  CLPGPU_DEBUG_STATS=1 timeout -k 5 120 python tools/flipdbg.py flip_scatter=$v pivots=4000 > gpurun_out/c12_dbg4k_v$v.log 2>&1
done
grep -h "clpgpu dbg\|pivots\|k_" gpurun_out/c12_dbg*.log | cut -c1-600
timeout -k 5 200 python tools/variant_check.py flip_scatter=2 flip_slot_cap=2 > gpurun_out/c12_variant.log 2>&1
timeout -k 5 300 python -m pytest tests -m gpu -x -q -k "flip or config4_full" > gpurun_out/c12_tests.log 2>&1
for v in 1 0; do
  CLPGPU_OPTS="flip_scatter=$v" timeout -k 5 120 python bench.py --steps 2000 --warmup 200 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c12_bench_v$v.log 2>&1
  CLPGPU_OPTS="flip_scatter=$v" timeout -k 5 100 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c12_driver_v$v.log 2>&1
done
grep identical gpurun_out/c12_variant.log; grep -E "passed|failed" gpurun_out/c12_tests.log | tail -3
for f in gpurun_out/c12_bench_v1.log gpurun_out/c12_bench_v0.log gpurun_out/c12_driver_v1.log gpurun_out/c12_driver_v0.log; do python -c "
import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); k=d['roofline']['per_kernel_us']; print('$f', round(d['value'],1), k.get('k_flip_apply2'), k.get('k_dj_flags'))"; done
