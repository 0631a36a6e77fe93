#!/bin/bash
# MFMA utilisation of the re-inversion's trailing update and of the engine's f64 GEMM (GPU box):
#   tools/pmc_mfma.sh <out file under gpurun_out/> <what, for the header> -- <command ...>
# e.g. tools/pmc_mfma.sh r06_mfma_sparse.txt "8000 mature pivots of config 4" -- python tools/refactor_probe.py "" 8000
# (counters in their own pass, --kernel-trace only, as the guide prescribes)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$1; WHAT=$2; shift 3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
(cd $R && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --kernel-include-regex "k_gj2_trail_mfma|k_dgemm" -d /tmp/pmc_mfma -o p -- "$@" > /tmp/pmc_mfma.log 2>&1)
f=$(find /tmp/pmc_mfma -name "*results.db" | head -1)
python - "$f" "$WHAT" "$*" > $R/gpurun_out/$OUT <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection group by kernel_name, counter_name").fetchall()
agg = {}
for k, c, n, s, a in rows:
    agg.setdefault(k.split('(')[0], {})[c] = (n, s, a)
print(f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -- {sys.argv[3]}  ({sys.argv[2]})")
for k, d in agg.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        n, busy, _ = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        _, act, _ = d["GRBM_GUI_ACTIVE"]
        # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs, GRBM_GUI_ACTIVE the 8 XCDs (profiles/r02_mfma_trail.txt): utilisation = busy / (1024 x active / 8)
        print(f"{k:50s} dispatches {n:6d}  MFMA busy cycles {busy:.4g}  GUI active cycles (8 XCDs) {act:.4g}  MFMA utilisation {100 * busy / (128.0 * act):.1f} %")
PY
cat $R/gpurun_out/$OUT
