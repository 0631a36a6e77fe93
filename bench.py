#!/usr/bin/env python3
"""bench.py -- dual-simplex iterations/sec of the HIP engine on the north-star workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one simplex pivot (one pass of the hot path: CHUZR, BTRAN, row pricing + ratio test,
FTRAN x2, dual/primal/weight updates, basis update).  Workload: BASELINE.json configs[3], the
synthetic 50 000 x 200 000 sparse LP (~10 M nonzeros, steepest-edge dual, slack start); inputs are
resident in HBM before the timed region.  With N > 1 the structural columns are priced in N
contiguous ranges, one per rank/GPU, and the per-rank tableau-row slices are exchanged with RCCL
(multi-GPU); every rank performs the same pivots, so `value` = pivots / max-over-ranks
time ("strong" scaling: the LP is fixed).

Prints ONE JSON line (rank 0) with the driver's fields plus
  roofline     -- the dominant kernel (row pricing, HBM-bound): algorithmic bytes per launch
                  (SURVEY.md 8d formula, counted by the kernel itself) / mean launch duration
                  measured with HIP events on the engine's stream, against 8 TB/s
  cpu_baseline -- the CPU oracle (a port of the reference loop; the reference itself cannot be built
                  without CoinUtils) on the same LP for a bounded number of pivots, 1 core
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides: ~6.3 TB/s achievable)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--rows", type=int, default=50000)
    ap.add_argument("--cols", type=int, default=200000)
    ap.add_argument("--nnz-per-col", type=int, default=50)
    ap.add_argument("--pivot-rule", type=int, default=1)
    ap.add_argument("--cpu-iterations", type=int, default=-1, help="pivots for the CPU baseline (-1 auto, 0 skip)")
    ap.add_argument("--check-every", type=int, default=16)
    ap.add_argument("--workload", default="sparse", choices=["sparse", "dense", "netlib"],
                    help="sparse = BASELINE configs[3] (default, the quoted metric); dense = configs[2]; netlib = power-law variant")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")

    from clp_amd import problems as P
    from clp_amd.engine import ClpGpuSimplex

    t0 = time.time()
    if args.workload == "dense":
        lp = P.dense_lp(args.rows, args.cols)
    elif args.workload == "netlib":
        lp = P.netlib_shaped_lp(args.rows, args.cols, args.rows * args.cols // 1000)
    else:
        lp = P.sparse_lp(args.rows, args.cols, args.nnz_per_col)
    gen_s = time.time() - t0
    eng = ClpGpuSimplex(local_rank).loadProblem(lp)
    eng.set_option("pivot_rule", args.pivot_rule)
    eng.set_option("check_every", args.check_every)
    # refactorization frequency as ClpSimplex::initialSolve sets it (defaultFactorizationFrequency:
    # 475 at m = 50 000); the CPU baseline below uses the same value
    eng.set_option("max_pivots", 0)
    if os.environ.get("CLPGPU_PRICE_KERNEL"):
        eng.set_option("price_kernel", int(os.environ["CLPGPU_PRICE_KERNEL"]))
    for kv in filter(None, os.environ.get("CLPGPU_OPTS", "").split(",")):  # experiment knobs, e.g. max_pivots=475
        key, val = kv.split("=")
        eng.set_option(key, float(val))
    if distributed or os.environ.get("CLPGPU_FORCE_COMM"):
        from clp_amd.multigpu import attach_communicator

        attach_communicator(eng, rank, world)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup: startup (factorize, resync) + W pivots, untimed
    status = eng.dual_steps(args.warmup)
    assert status == -1, f"LP finished during warmup (status {status})"
    it0 = eng.numberIterations()
    barrier()
    t1 = time.perf_counter()
    status = eng.dual_steps(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    barrier()
    steps_done = eng.numberIterations() - it0
    assert steps_done == args.steps, f"timed {steps_done} pivots, wanted {args.steps} (status {status})"
    # kernel-level timing of the dominant kernel: HIP events around every pricing launch on the
    # engine's stream, over the pivots that immediately follow the timed region (event records are
    # not replayable inside the hipGraph the timed region uses, so this leg launches eagerly)
    eng.set_option("timing", 1)
    s0 = eng.stats()
    eng.dual_steps(min(args.steps, 500))
    torch.cuda.synchronize()
    s1 = eng.stats()
    eng.set_option("timing", 0)
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    launches = s1["price_launches"] - s0["price_launches"]
    price_ms = s1["price_ms"] - s0["price_ms"]
    price_bytes = s1["price_bytes"] - s0["price_bytes"]
    per_launch_bytes = price_bytes / max(launches, 1)
    per_launch_s = price_ms * 1e-3 / max(launches, 1)
    achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0

    # HBM traffic of the same kernel from PMC counters: collected in separate rocprofv3 --pmc passes
    # (profiles/r01_pmc_price_sell.txt) -- counters cannot be read from inside this process
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_price_sell.json")
    if os.path.exists(pmc_path) and (args.workload, args.rows, args.cols, args.nnz_per_col) == ("sparse", 50000, 200000, 50) and world == 1:
        traffic = json.load(open(pmc_path))["traffic_bytes_per_launch"]

    cpu = None
    if rank == 0 and args.cpu_iterations != 0:
        from oracle.oracle import OracleSimplex

        o = OracleSimplex(lp)
        o.set_option("pivot_rule", args.pivot_rule)
        o.set_option("max_pivots", 0)
        n_cpu = args.cpu_iterations if args.cpu_iterations > 0 else args.warmup + args.steps
        o.set_option("max_iterations", n_cpu)
        o.dual()
        cpu = {"value": o.iterations / max(o.seconds, 1e-9), "unit": "iterations/s", "cores": 1, "kind": "port",
               "sample": f"first {o.iterations} pivots of the same LP from the slack basis ({o.seconds:.1f} s), "
                         "CPU oracle = C restatement of ClpSimplexDual (reference needs CoinUtils, not buildable here)"}

    config_ref = {"sparse": "BASELINE.json configs[3]", "dense": "BASELINE.json configs[2]",
                  "netlib": "Netlib-shaped variant of BASELINE.json configs[3]"}[args.workload]
    if (args.rows, args.cols) != {"dense": (5000, 5000)}.get(args.workload, (50000, 200000)):
        config_ref += ", non-default size"
    # mean column length >= 256 selects the wave-per-column pricing kernel (engine.hip, widePricing)
    price_kernel = ("k_price_wide (row pricing by column, a wave per column, fused first ratio pass)"
                    if len(lp.elem) >= 256 * lp.n else
                    "k_price_sell (row pricing by column, SELL-64 + a workgroup per long column, fused first ratio pass)")
    if rank == 0:
        out = {
            "metric": "dual-simplex iterations/sec",
            "value": args.steps / elapsed,
            "unit": "iterations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.workload} LP {lp.m}x{lp.n}, {len(lp.elem)} nonzeros ({config_ref}), "
                                   + ("steepest-edge dual" if args.pivot_rule else "Dantzig dual") + " from the slack basis",
                       "rows": int(lp.m), "cols": int(lp.n), "nnz": int(len(lp.elem)),
                       "parallelism": f"column-range pricing x{world}" if world > 1 else "1 GPU",
                       "check_every": args.check_every, "generate_s": round(gen_s, 1)},
            "roofline": {"bound": "hbm", "kernel": price_kernel,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_launch": per_launch_bytes, "us_per_launch": per_launch_s * 1e6,
                         "launches": int(launches), "traffic": traffic},
            "cpu_baseline": cpu,
            "refactorizations": int(s1["refactorizations"]),
        }
        print(json.dumps(out))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
