#!/bin/bash
# rocprofv3 kernel stats of a solve stretch on the GPU box:  tools/prof_solve.sh <tag> <budget s> [opts] [workload]
# writes gpurun_out/r03_prof_<tag>.jsonl (solve profile) and gpurun_out/r03_prof_<tag>_kernel_stats.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/prof_$1
rocprofv3 --kernel-trace --stats -d /tmp/prof_$1 -o p -- python $R/tools/solve_profile.py --workload ${4:-sparse} --budget $2 --chunk 4000 --opts "$3" > $R/gpurun_out/r03_prof_$1.jsonl 2> $R/gpurun_out/r03_prof_$1.err
f=$(find /tmp/prof_$1 -name "*results.db" | head -1)
python $R/tools/rocpd_summary.py "$f" "config-4 solve stretch, $2 s, opts [$3]" | head -50 > $R/gpurun_out/r03_prof_$1_kernel_stats.txt
