#!/bin/bash
# config-4 solve stretches under different LU settings (GPU box): tools/tune_lu.sh <budget s> "<opts1>" "<opts2>" ...
R=${GRAFT_REPO_ROOT:-/root/repo}
b=$1; shift
i=0
for o in "$@"; do
  i=$((i+1))
  python $R/tools/solve_profile.py --workload sparse --budget $b --chunk 4000 --opts "$o" > $R/gpurun_out/r03_tune_$i.jsonl 2>/dev/null
  echo "== [$o]"; tail -n 3 $R/gpurun_out/r03_tune_$i.jsonl | head -n 2 | cut -c1-260
done
