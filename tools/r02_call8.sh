#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c8_build.log 2>&1
timeout -k 5 200 python tools/variant_check.py dc_variant=1 > gpurun_out/c8_variant.log 2>&1
for v in 0 1; do
  CLPGPU_OPTS="dc_variant=$v" timeout -k 5 120 python bench.py --steps 2000 --warmup 200 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c8_bench_v$v.log 2>&1
  CLPGPU_OPTS="dc_variant=$v" timeout -k 5 100 python bench.py --steps 20 --warmup 5 --tto-budget 0 --pmc off --cpu-iterations 0 > gpurun_out/c8_driver_v$v.log 2>&1
done
cat gpurun_out/c8_variant.log | tail -5
for f in gpurun_out/c8_bench_v0.log gpurun_out/c8_bench_v1.log gpurun_out/c8_driver_v0.log gpurun_out/c8_driver_v1.log; do python -c "
import json,sys; d=json.loads([l for l in open('$f') if l.startswith('{')][-1]); print('$f', round(d['value'],1), d['roofline']['per_kernel_us']['k_dual_column'], d['roofline']['per_kernel_us']['k_fix_house'])"; done
