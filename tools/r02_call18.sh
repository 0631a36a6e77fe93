#!/bin/bash
# config 4 to optimality (or to the budget) with the default engine
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c18_build.log 2>&1
timeout -k 5 $(( ${1:-330} + 60 )) python tools/solve_profile.py --workload sparse --budget ${1:-330} --chunk 4000 > gpurun_out/c18_solve.log 2>&1
python - <<'PY'
import json
rows=[json.loads(l) for l in open('gpurun_out/c18_solve.log') if l.startswith('{')]
for r in rows[:-1][::5]: print(r['iterations'], r['elapsed_s'], r['chunk_it_per_s'], r['nucleus'], r['refactorizations'], r['refreshes'], r['refreshes_rejected'], round(r['objective'],1))
print(rows[-1])
PY
