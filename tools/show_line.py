"""prints the parts of a bench line a builder looks at first (lab tool): python tools/show_line.py gpurun_out/bench.json"""
import json
import sys

d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
print("value", d["value"], "ms", d["ms_per_step"])
r = d["roofline"]
print({k: r.get(k) for k in ("kernel", "frac", "us_per_launch", "duration_source", "kernel_trace", "traffic")})
print(r.get("eager_events"))
c = d["cpu_baseline"]
if c:
    print({k: c[k] for k in c if k not in ("slack_start", "sample")})
    if c.get("slack_start"):
        print("slack", {k: c["slack_start"][k] for k in ("value", "window", "seconds")}, c["slack_start"]["gpu_same_window"])
print("tto", d["time_to_optimal"])
if d.get("sustained"):
    print("sustained", d["sustained"]["windows"], d["sustained"]["over_the_whole_leg"])
if d.get("time_to_optimal_ladder"):
    for r_ in d["time_to_optimal_ladder"]["rungs"]:
        print({k: r_.get(k) for k in ("rung", "status", "engine_seconds", "engine_iterations", "objective_matches_highs", "certified_optimal", "skipped")})
print("sub", json.dumps(d.get("sub_records"))[:3000])
print("shard", json.dumps(d.get("shard_pricing_proxy")))
print("causes", d.get("status_check_causes"))
if d.get("roofline_mature"):
    for k in d["roofline_mature"]["kernels"]:
        print("   ", {x: k.get(x) for x in ("kernel", "us_per_launch", "launches_per_pivot", "share_of_kernel_time", "frac")})
