#!/bin/bash
# lanes-per-column variants of the pricing layout (option sell_lanes): pivots/s, us per pricing launch, GB/s
for L in 1 2 4 8; do
  echo "sell_lanes $L"
  CLPGPU_OPTS="sell_lanes=$L" timeout 200 python bench.py --steps 2500 --warmup 200 --cpu-iterations 0 2>&1 | tail -1 |
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['us_per_launch'], d['roofline']['achieved'])"
done
# and parity of the priced row against the default layout on a small LP
python - <<'PY'
import numpy as np
from clp_amd import problems as P
from clp_amd.engine import ClpGpuSimplex
lp = P.sparse_lp(1500, 6000, 10, 31)
ref = None
for L in (1, 2, 4, 8):
    g = ClpGpuSimplex().loadProblem(lp)
    g.set_option("sell_lanes", L)
    st = g.dual()
    print("sell_lanes", L, "status", st, "iterations", g.numberIterations(), "objective", g.objectiveValue())
PY
