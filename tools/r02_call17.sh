#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c17_build.log 2>&1
timeout -k 5 300 python -m pytest tests -m gpu -x -q -k "verified_refresh" > gpurun_out/c17_tests.log 2>&1
grep -E "passed|failed|Error|assert" gpurun_out/c17_tests.log | tail -8
timeout -k 5 200 python tools/solve_profile.py --workload sparse --budget ${1:-60} --chunk 2000 --opts refresh_min_k=6144,refresh_tolerance=2e-6,log_level=2 > gpurun_out/c17_solve.log 2>&1
grep "clpgpu:" gpurun_out/c17_solve.log | tail -24
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/c17_solve.log') if l.startswith('{')]
for r in rows[:-1][::3]: print(r['iterations'], r['elapsed_s'], r['chunk_it_per_s'], r['nucleus'], r['refactorizations'], r['refreshes'], r['refreshes_rejected'], round(r['objective'],1))
print(rows[-1])
PY
