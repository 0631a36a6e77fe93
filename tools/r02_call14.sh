#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/c14_build.log 2>&1
(cd /tmp && timeout -k 5 200 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/c14_prof" -o run -- python "$GRAFT_REPO_ROOT/tools/solve_profile.py" --workload sparse --budget 45 --chunk 2000 > "$GRAFT_REPO_ROOT/gpurun_out/c14_solve.log" 2>&1)
python tools/rocpd_summary.py $(find gpurun_out/c14_prof -name "*_results.db" | head -1) "round 2, config 4 first 45 s of the solve (nucleus grows to ~10 000): rocprofv3 --kernel-trace --stats -- python tools/solve_profile.py --workload sparse --budget 45" > gpurun_out/c14_kernel_stats.txt 2>&1
# the same, late window only (kernels after the first 25 s)
python - <<'PY' > gpurun_out/c14_late.txt 2>&1
import sqlite3, glob
db = glob.glob("gpurun_out/c14_prof/**/*_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
t0, t1 = cur.execute("select min(start), max(end) from kernels").fetchone()
cut = t1 - 12e9
rows = cur.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, max(end-start)/1e3 from kernels where start > ? group by name order by 3 desc limit 40", (cut,)).fetchall()
tot = sum(r[2] for r in rows)
print("last 12 s of the trace; total kernel time us", tot)
for r in rows:
    print(f"{r[0][:70]:70s} {r[1]:8d} {r[2]:12.0f} {r[3]:10.2f} {r[4]:10.2f} {100*r[2]/tot:6.2f}")
PY
rm -rf gpurun_out/c14_prof
tail -3 gpurun_out/c14_solve.log | cut -c1-300; head -45 gpurun_out/c14_late.txt
