run() { echo "== $1"; CLPGPU_OPTS="$1" timeout 120 python bench.py --steps 2000 --warmup 200 --cpu-iterations 0 --check-every ${2:-16} 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), 'it/s', round(1e3*d['ms_per_step']), 'us/it; price', round(d['roofline']['us_per_launch'],1), 'us', d['refactorizations'], 'refactors')"; }
run ""
run "use_graph=0"
run "max_pivots=475"
run "max_pivots=2000"
run "" 64
run "max_pivots=2000" 64
