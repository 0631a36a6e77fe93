#!/bin/bash
# HBM read traffic (FETCH_SIZE, KiB; x 2 on gfx950 for streaming reads, see bench.py) of the streaming kernels of a mature LU-mode pivot,
# counters in their own pass with --kernel-trace only:  tools/pmc_streams.sh -> gpurun_out/r06_pmc_streams.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_streams
rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "k_lu_gemv3|k_lu_gemvT|k_lu_eta_apply|k_price_lds|k_ftran_scatter3_lu" -d /tmp/pmc_streams -o p -- python $R/bench.py --pmc-child --steps 300 --warmup 100 > /tmp/pmc_streams.log 2>&1
f=$(find /tmp/pmc_streams -name "*results.db" | head -1)
python - "$f" > $R/gpurun_out/r06_pmc_streams.txt <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
print("rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python bench.py --pmc-child --steps 300 --warmup 100  (config 4 from the mature basis: 800 pre-roll + 400 eager pivots; last 300 dispatches of each kernel)")
for name in ("k_lu_gemv3", "k_lu_gemvT", "k_lu_eta_apply", "k_ftran_scatter3_lu", "k_price_lds"):
    rows = cur.execute("select value from counters_collection where kernel_name like ? and counter_name = 'FETCH_SIZE' order by dispatch_id desc limit 300", (f"%{name}(%",)).fetchall()
    if rows:
        v = [r[0] for r in rows]
        print(f"{name:24s} dispatches {len(v):4d}  FETCH_SIZE mean {sum(v) / len(v):12.1f} KiB  -> read {2.0 * 1024.0 * sum(v) / len(v) / 1e6:8.1f} MB per launch (2 x FETCH_SIZE)")
PY
cat $R/gpurun_out/r06_pmc_streams.txt
