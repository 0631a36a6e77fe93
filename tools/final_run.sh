#!/bin/bash
# The round's closing GPU run (on the GPU box):  tools/final_run.sh [tag]
# full -m gpu suite, the bench line (default and driver-sized window), rocprofv3 kernel stats of the bench command, the bench with a
# forced one-rank communicator (the sharded chain, LU mode), the pricing probe.  Everything lands in gpurun_out/<tag>_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
T=${1:-r06}
O=$R/gpurun_out
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q --durations=25 > $O/${T}_gpu_suite_final.txt 2>&1
tail -4 $O/${T}_gpu_suite_final.txt
python -m pytest tests/test_gpu_mature_parity.py tests/test_gpu_chuzr.py -q -s 2>&1 | grep -v "^$" | tail -40 > $O/${T}_mature_parity_and_chuzr.txt
python bench.py > $O/${T}_bench_line_default.json 2> $O/${T}_bench_line_default.err
python bench.py --steps 20 --warmup 5 --ladder-budget 0 --tto-budget 5 > $O/${T}_bench_line_driver_window.json 2> $O/${T}_bench_line_driver_window.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$T
rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o p -- python $R/bench.py --cpu-iterations 0 --pmc off --tto-budget 0 --ladder-budget 0 --sub-records off --shard-proxy off > $O/${T}_bench_under_rocprof.json 2> $O/${T}_bench_under_rocprof.err
f=$(find /tmp/prof_$T -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py "$f" "rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 --ladder-budget 0 --sub-records off --shard-proxy off (headline + slack-start leg + eager replay, config 4 from the mature basis)" | head -60 > $O/${T}_bench_kernel_stats.txt
head -30 $O/${T}_bench_kernel_stats.txt
# the mature stretch alone (800 pre-roll + 100 + 600 eager pivots from the committed basis): per-kernel durations as CSV
rm -rf /tmp/prof_${T}_m
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T}_m -o m -- python $R/bench.py --pmc-child --steps 600 --warmup 100 > /dev/null 2>&1 < /dev/null
find /tmp/prof_${T}_m -name "*kernel_stats.csv" -exec cp {} $O/${T}_mature_stretch_kernel_stats_final.csv \;
grep -E "price_lds|dual_column|gemv3|gemvT|scatter3_lu" $O/${T}_mature_stretch_kernel_stats_final.csv | cut -d, -f1-4
cd $R
python tools/eta_probe.py lu_compact_eta=1 lu_compact_eta=0 2>&1 | grep -v amdgpu.ids > $O/${T}_eta_probe.txt
tools/pmc_mfma.sh ${T}_mfma_dense.txt "BASELINE configs[2]: dense 5000 x 5000 to optimality" -- python bench.py --workload dense --rows 5000 --cols 5000 --steps 300 --warmup 100 --tto-budget 30 --cpu-iterations 0 --pmc off --ladder-budget 0 --sub-records off --shard-proxy off --start slack > /dev/null 2>&1
tools/pmc_mfma.sh ${T}_mfma_sparse.txt "8000 pivots of config 4 from the mature basis: the tail inversions of its refactorizations" -- python tools/refactor_probe.py "" 8000 > /dev/null 2>&1
for f in default driver_window; do python - <<PY
import json
d=json.loads(open("$O/${T}_bench_line_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],3), "us", round(d["roofline"]["us_per_launch"],1), "moved_frac", d["roofline"].get("moved_frac"))
PY
done
