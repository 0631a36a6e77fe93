#!/bin/bash
# Objective against time on config 4 for different resync intervals, on the GPU box:
#   tools/objective_race.sh [budget s, default 40]
# One solve stretch per setting, each with --theta-stats (share of pivots whose dual step is below 1e-9, bound flips per pivot);
# writes gpurun_out/race_<tag>.jsonl and prints, per setting, the objective reached at the end of the budget.
R=${GRAFT_REPO_ROOT:-/root/repo}
b=${1:-40}
mkdir -p $R/gpurun_out
run() {  # tag, options
  python $R/tools/solve_profile.py --workload sparse --budget $b --chunk 4000 --theta-stats --opts "$2" > $R/gpurun_out/race_$1.jsonl 2> $R/gpurun_out/race_$1.err
  echo "== $1 [$2]"; tail -n 2 $R/gpurun_out/race_$1.jsonl | head -n 1 | cut -c1-330
}
run lu_adaptive ""
run lu_475 "lu_max_pivots=475"
run lu_200 "lu_max_pivots=200,lu_min_pivots=100"
run explicit_inverse "factor_mode=0"
