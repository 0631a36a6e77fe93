#!/bin/bash
# round 2, GPU call 5: everything green?  then the numbers the docs quote.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c5_build.log 2>&1
timeout -k 5 900 python -m pytest tests -m gpu -q -rf --timeout 400 -p no:cacheprovider > gpurun_out/c5_pytest.log 2>&1
echo "pytest rc $?" >> gpurun_out/c5_pytest.log
timeout -k 5 420 python bench.py --steps 2000 --warmup 200 --tto-budget 40 > gpurun_out/c5_bench.log 2> gpurun_out/c5_bench.err
timeout -k 5 200 python bench.py --steps 20 --warmup 5 > gpurun_out/c5_bench_driver.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c5_prof" -o run -- python "$GRAFT_REPO_ROOT/bench.py" --cpu-iterations 0 --pmc off --tto-budget 0 > "$GRAFT_REPO_ROOT/gpurun_out/c5_prof.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c5_prof -name "*_results.db" | head -1) "round 2 final chain: rocprofv3 --kernel-trace --stats -- python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 (headline in hipGraph mode + two eager replay legs of the same 2200 pivots)" > gpurun_out/c5_kernel_stats.txt 2>&1
rm -rf gpurun_out/c5_prof
# dense config: kernel trace of a whole solve (re-inversion kernels) + MFMA counters of the trailing update
(cd /tmp && timeout -k 5 120 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c5_dprof" -o run -- python "$GRAFT_REPO_ROOT/tools/solve_profile.py" --workload dense --budget 30 --chunk 2000 > "$GRAFT_REPO_ROOT/gpurun_out/c5_dense_solve.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c5_dprof -name "*_results.db" | head -1) "round 2, dense 5000 x 5000 to optimality: rocprofv3 --kernel-trace --stats -- python tools/solve_profile.py --workload dense" > gpurun_out/c5_dense_kernel_stats.txt 2>&1
rm -rf gpurun_out/c5_dprof
(cd /tmp && timeout -k 5 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_gj2_trail_mfma" -d "$GRAFT_REPO_ROOT/gpurun_out/c5_mfma" -o run -- python "$GRAFT_REPO_ROOT/tools/solve_profile.py" --workload dense --budget 30 --chunk 2000 > "$GRAFT_REPO_ROOT/gpurun_out/c5_mfma.log" 2>&1)
python - <<'PY' > gpurun_out/c5_mfma_summary.txt 2>&1
import sqlite3, glob
db = glob.glob("gpurun_out/c5_mfma/**/*_results.db", recursive=True)
cur = sqlite3.connect(db[0]).cursor()
print("tables", [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")][:40])
for name in ("SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
    rows = cur.execute("select count(*), avg(value), min(value), max(value), sum(value) from counters_collection where counter_name = ?", (name,)).fetchall()
    print(name, rows)
rows = cur.execute("select count(*), sum(end-start)/1e3, avg(end-start)/1e3 from kernels where name like '%k_gj2_trail_mfma%'").fetchall()
print("k_gj2_trail_mfma dispatches, total us, avg us", rows)
PY
rm -rf gpurun_out/c5_mfma
tail -4 gpurun_out/c5_pytest.log; tail -c 1200 gpurun_out/c5_bench_driver.log
