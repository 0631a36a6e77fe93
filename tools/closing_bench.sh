R=${GRAFT_REPO_ROOT:-/root/repo}
T=r05
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 420 python bench.py > $O/${T}_bench_line_default.json 2> $O/${T}_bench_line_default.err < /dev/null
timeout 200 python bench.py --steps 20 --warmup 5 --ladder-budget 0 --tto-budget 5 > $O/${T}_bench_line_driver_window.json 2> $O/${T}_bench_line_driver_window.err < /dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$T
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o p -- python $R/bench.py --cpu-iterations 0 --pmc off --tto-budget 0 --ladder-budget 0 > $O/${T}_bench_under_rocprof.json 2> $O/${T}_bench_under_rocprof.err < /dev/null
f=$(find /tmp/prof_$T -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py "$f" "rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 --ladder-budget 0 (headline + slack-start leg + eager replay, config 4 from the mature basis)" 2>&1 | head -60 > $O/${T}_bench_kernel_stats.txt
head -16 $O/${T}_bench_kernel_stats.txt
rm -rf /tmp/prof_${T}_m
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${T}_m -o m -- python $R/bench.py --pmc-child --steps 600 --warmup 100 > /dev/null 2>&1 < /dev/null
find /tmp/prof_${T}_m -name "*kernel_stats.csv" -exec cp {} $O/${T}_mature_stretch_kernel_stats_final.csv \;
grep -E "price_lds|dual_column|gemv3|gemvT|scatter3_lu" $O/${T}_mature_stretch_kernel_stats_final.csv | cut -d, -f1-4
cd $R
for f in default driver_window; do python - <<PY
import json
d=json.loads(open("$O/${T}_bench_line_$f.json").read().strip().splitlines()[-1])
print("$f", round(d["value"],1), d["unit"], "frac", round(d["roofline"]["frac"],3), "us", round(d["roofline"]["us_per_launch"],1), "moved_frac", d["roofline"].get("moved_frac"))
PY
done
