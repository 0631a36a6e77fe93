#!/usr/bin/env python3
"""bench.py -- dual-simplex iterations/sec (+ time-to-optimal) of the HIP engine on the north-star workload.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one simplex pivot (one pass of the hot path: CHUZR, BTRAN, row pricing + ratio test,
FTRAN x2, dual/primal/weight updates, basis update).  Workload: BASELINE.json configs[3], the synthetic
50 000 x 200 000 sparse LP (~10 M nonzeros, steepest-edge dual, slack start, the reference's default
refactorization frequency); inputs are resident in HBM before the timed region.

Legs (rank 0 prints ONE JSON line):
  headline      the context is WARM-STARTED FROM A MATURE BASIS of the same LP (tests/golden/basis_sparse_30000.npy: the status
                array of this LP after 30 000 pivots, nucleus ~11 000, written by tools/dump_basis.py) -- the regime a solve of this
                LP spends nearly all of its time in: pi dense, pricing by column, LU mode.  W untimed warm-up pivots (the start-up
                factorization of that basis is part of them), then EXACTLY K pivots as captured hipGraphs, bracketed by barrier +
                synchronize; `value` = K / max-over-ranks time.  `slack_start` = the same count of pivots from the slack basis
                (the near-identity regime earlier rounds quoted as the headline), a secondary field.  --start slack restores it.
  roofline      the pricing kernel of the timed window.  `achieved` = algorithmic bytes (counted by the kernel) / the
                kernel's OWN average duration over the timed pivots, taken from a `rocprofv3 --kernel-trace` child pass of this script
                (`kernel_trace`, `duration_source`); the same bytes over the HIP-event time of an EAGER replay of the window (a second
                context: the engine is deterministic; event pair + launch gap included) stays as `eager_events`, and is what `frac`
                falls back to when rocprofv3 cannot run.  `per_kernel_us` = every kernel of the chain over the same replay.
                `traffic` = HBM bytes per pricing launch from PMC counters: two more child passes (`--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`,
                each with --kernel-trace only), mean of the launches of the timed window; falls back to the committed summary under
                profiles/ (named in `traffic_source`).  `moved_frac` = traffic / time / peak.
  cpu_baseline  the CPU oracle (a C restatement of the reference loop -- the reference needs CoinUtils and cannot be built here) LIVE
                on this box IN THE HEADLINE'S REGIME: warm-started from the same committed mature basis for --cpu-mature-pivots
                pivots (40: ~10 s of one core), its start-up factorization (a dense LU of order 10 514, ~15 s, threaded) clocked apart
                and excluded on both sides; `gpu_same_window` = the engine over exactly those pivots after its own start-up
                factorization, with the count of identical pivots (both sides under ClpDualRowSteepest's full scan in this leg: see
                the comment at the leg).  `slack_start` = the pair of rounds 1-5 (oracle and engine over pivots 1..2000 from the
                slack basis).  `clp_upstream` = real `clp` on the same LP written as MPS when a clp binary is on PATH; null otherwise.
  sub_records   BASELINE configs[2] (dense 5000 x 5000) and the Netlib-shaped variant, each a child run of this script: window from
                the slack basis, pricing roofline, time to optimal.
  shard_pricing_proxy  the pricing kernel of a column shard [0, n / R) for R = 1, 2, 4, 8 timed on this one GPU (no 8-GPU node exists
                for this repository): what a rank of an R-GPU run launches per pivot.
  time_to_optimal  the solve continued to optimality within --tto-budget seconds (or how far it got); compared with
                the independent optimum under tests/golden/bench_optima.json when there is one.
  time_to_optimal_ladder  LPs of the same generator at sizes an independent solver finishes (tools/ladder.py, optima in
                tests/golden/ladder_optima.json): the engine in its default mode from the slack basis to status 0, seconds and
                iterations next to HiGHS' (one core) and the oracle port's, objective against HiGHS' to 1e-8; rungs in ascending
                size while --ladder-budget seconds last.
  sustained     pivots/s of that continuation in windows of 2000 pivots: the first, the one where the nucleus passes
                5000, and the last reached -- refactorizations included.
  roofline_mature  per-kernel time, algorithmic bytes and HBM fraction of the kernels that dominate the mature
                regime (dense tail GEMVs, eta file, pricing by column with dense pi, ratio test).
"""
import argparse
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guide: ~6.3 TB/s achievable with a float4 copy)


def make_lp(args):
    from clp_amd import problems as P

    if args.workload == "dense":
        return P.dense_lp(args.rows, args.cols)
    if args.workload == "netlib":
        return P.netlib_shaped_lp(args.rows, args.cols, args.rows * args.cols // 1000)
    return P.sparse_lp(args.rows, args.cols, args.nnz_per_col)


MATURE_BASIS = os.path.join(ROOT, "tests", "golden", "basis_sparse_30000.npy")


def mature_basis(args):
    """status array of the default bench LP after 30 000 pivots (tools/dump_basis.py sparse 30000), or None"""
    default = (args.workload, args.rows, args.cols, args.nnz_per_col) == ("sparse", 50000, 200000, 50)
    if args.start != "mature" or not default or not os.path.exists(MATURE_BASIS):
        return None
    import numpy as np

    return np.load(MATURE_BASIS).astype(np.uint8)


def make_engine(args, lp, device, basis=None):
    from clp_amd.engine import ClpGpuSimplex

    eng = ClpGpuSimplex(device).loadProblem(lp)
    if basis is not None:
        eng.setStatusArray(basis)
    eng.set_option("pivot_rule", args.pivot_rule)
    eng.set_option("check_every", args.check_every)
    # refactorization frequency as ClpSimplex::initialSolve sets it (defaultFactorizationFrequency:
    # 475 at m = 50 000); the CPU baseline uses the same value
    eng.set_option("max_pivots", 0)
    if os.environ.get("CLPGPU_PRICE_KERNEL"):
        eng.set_option("price_kernel", int(os.environ["CLPGPU_PRICE_KERNEL"]))
    for kv in filter(None, os.environ.get("CLPGPU_OPTS", "").split(",")):  # experiment knobs, e.g. max_pivots=475
        key, val = kv.split("=")
        eng.set_option(key, float(val))
    return eng


def pmc_traffic(args, kernel_regex):
    """HBM bytes per pricing launch over the timed window: two rocprofv3 --pmc passes of this script
    (child mode: warm-up + steps, nothing else).  Returns (bytes, note) or (None, why)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="clpgpu_pmc_")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--kernel-include-regex", kernel_regex, "-d", d, "-o", "pmc", "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--rows", str(args.rows), "--cols", str(args.cols), "--nnz-per-col", str(args.nnz_per_col), "--workload", args.workload,
               "--pivot-rule", str(args.pivot_rule), "--check-every", str(args.check_every), "--start", args.start, "--preroll", str(args.preroll)]
        try:
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
            try:
                p.communicate(timeout=args.pmc_timeout)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, 9)
                p.communicate()
                return None, f"rocprofv3 --pmc {counter} timed out"
            dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
            if p.returncode != 0 or not dbs:
                return None, f"rocprofv3 --pmc {counter} failed (rc {p.returncode})"
            cur = sqlite3.connect(dbs[0]).cursor()
            rows = cur.execute("select value from counters_collection where counter_name = ? order by dispatch_id desc limit ?",
                               (counter, args.steps)).fetchall()
            if not rows:
                return None, f"no {counter} rows for {kernel_regex}"
            out[counter] = sum(r[0] for r in rows) / len(rows)
        except Exception as e:  # noqa: BLE001 -- the bench line must survive a profiler problem
            return None, f"{counter}: {e}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    # guide: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per 128-B request of
    # a streaming read -> x2 (self-calibrated on the unconditional-stream kernel, profiles/r01_pmc_price_sell_v5.txt:
    # 125.5 MB read vs 120.5 MB algorithmic); WRITE_SIZE taken as is
    return 2.0 * out["FETCH_SIZE"] * 1024.0 + out["WRITE_SIZE"] * 1024.0, \
        f"live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, mean of the last {args.steps} pricing dispatches, read = 2 x FETCH_SIZE"


def kernel_trace_us(args, kernel_regex):
    """the pricing kernel's OWN duration over the timed window: one `rocprofv3 --kernel-trace` pass of this script (the same child
    as the PMC passes: warm-up + steps, eager launches, nothing else), mean of end - start over the last K dispatches.
    Returns ({"avg_us", "min_us", "max_us", "dispatches", "kernel"}, note) or (None, why)."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    d = tempfile.mkdtemp(prefix="clpgpu_kt_")
    cmd = [exe, "--kernel-trace", "--kernel-include-regex", kernel_regex, "-d", d, "-o", "kt", "--",
           sys.executable, os.path.abspath(__file__), "--pmc-child", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--rows", str(args.rows), "--cols", str(args.cols), "--nnz-per-col", str(args.nnz_per_col), "--workload", args.workload,
           "--pivot-rule", str(args.pivot_rule), "--check-every", str(args.check_every), "--start", args.start, "--preroll", str(args.preroll)]
    try:
        p = subprocess.Popen(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, start_new_session=True)
        try:
            p.communicate(timeout=args.pmc_timeout)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            p.communicate()
            return None, "rocprofv3 --kernel-trace timed out"
        dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith("_results.db")]
        if p.returncode != 0 or not dbs:
            return None, f"rocprofv3 --kernel-trace failed (rc {p.returncode})"
        cur = sqlite3.connect(dbs[0]).cursor()
        rows = cur.execute("select name, (end - start) / 1e3 from kernels where name like ? order by start desc limit ?",
                           (f"%{kernel_regex}%", args.steps)).fetchall()
        if not rows:
            return None, f"no dispatches of {kernel_regex}"
        us = [r[1] for r in rows]
        names = sorted({re.sub(r"\(.*", "", r[0]).replace("clpgpu::", "") for r in rows})
        return {"avg_us": sum(us) / len(us), "min_us": min(us), "max_us": max(us), "dispatches": len(us), "kernel": " + ".join(names)}, \
            f"live: rocprofv3 --kernel-trace child pass of this script, mean of end - start over the last {len(us)} pricing dispatches (the timed pivots, eager launches)"
    except Exception as e:  # noqa: BLE001 -- the bench line must survive a profiler problem
        return None, f"kernel trace: {e}"
    finally:
        shutil.rmtree(d, ignore_errors=True)


def clp_upstream(args, lp):
    """real coin-or/Clp on the same LP, when a clp binary exists on this box"""
    exe = shutil.which("clp")
    if not exe:
        return None
    from clp_amd.mps import write_mps

    path = os.path.join(tempfile.gettempdir(), f"clpgpu_bench_{os.getpid()}.mps")
    try:
        write_mps(lp, path)
        cmd = [exe, path, "-presolve", "off", "-scaling", "off", "-perturb", "off", "-dualpivot", "steepest", "-dualsimplex"]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.clp_timeout)
        wall = time.perf_counter() - t0
        m = re.search(r"(\d+)\s+iterations\s+time\s+([0-9.eE+-]+)", r.stdout)  # doc/clp-output-comparison.md:14
        if not m:
            return {"error": "could not parse clp output", "tail": r.stdout[-300:]}
        its, secs = int(m.group(1)), float(m.group(2))
        return {"value": its / max(secs, 1e-9), "unit": "iterations/s", "iterations": its, "seconds": secs, "wall_s": wall,
                "cores": 1, "kind": "reference", "command": " ".join(cmd[:1] + ["<lp>.mps"] + cmd[2:])}
    except subprocess.TimeoutExpired:
        return {"error": f"clp did not finish in {args.clp_timeout} s"}
    finally:
        if os.path.exists(path):
            os.remove(path)


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
    ap.add_argument("--tto-budget", type=float, default=12.0, help="seconds allowed for the time-to-optimal leg (0 skips it)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "off"], help="live PMC traffic of the pricing kernel via rocprofv3 child runs")
    ap.add_argument("--pmc-timeout", type=float, default=150.0)
    ap.add_argument("--clp-timeout", type=float, default=600.0)
    ap.add_argument("--start", default="mature", choices=["mature", "slack"],
                    help="mature (default): the timed context starts from the committed basis of the default LP after 30 000 pivots; "
                         "slack: from the slack basis (other workloads always do)")
    ap.add_argument("--preroll", type=int, default=800,
                    help="mature start only: untimed pivots between the start-up factorization of the committed basis and the W warm-up pivots, "
                         "so that the timed window sits in the middle of an eta-file cycle (its length runs 0 .. ~1600) and not right "
                         "behind a fresh factorization, where pivots are at their cheapest")
    ap.add_argument("--ladder-budget", type=float, default=60.0, help="seconds allowed for the time-to-optimal ladder (0 skips it)")
    ap.add_argument("--ladder-rungs", default="1500,3000,5000,7000",
                    help="rungs of tools/ladder.py's table to solve (also there: 2000, 4000, 10000 -- 10000 does not finish)")
    ap.add_argument("--cpu-mature-pivots", type=int, default=30,
                    help="pivots of the CPU baseline in the headline's regime (oracle from the mature basis, ~0.4 s each after a ~15 s start-up; 0 skips it)")
    ap.add_argument("--sub-timeout", type=float, default=240.0)
    ap.add_argument("--sub-records", default="dense,netlib",
                    help="default workload only: BASELINE configs[2] (dense 5000 x 5000) and the Netlib-shaped variant as sub-records of the line, "
                         "each a child run of this script (window from the slack basis, pricing roofline, time to optimal); 'off' skips them")
    ap.add_argument("--shard-proxy", default="1,2,4,8",
                    help="default workload only: the pricing kernel of a column shard [0, n / R) timed on this one GPU for each R "
                         "(what a rank of an R-GPU run launches per pivot); 'off' skips it")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if args.gpus != world and not args.pmc_child:
        # one process per GPU: `--gpus N` is what the launcher was asked for, WORLD_SIZE what it provided.  A bare
        # `python bench.py --gpus 8` used to run on one GPU and print n_gpus 1: refuse instead.
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE is {world}: start it as\n  python -m torch.distributed.run --nnodes=1 "
                         f"--nproc-per-node {args.gpus} --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus {args.gpus} ...")
    if distributed:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP engine has no CPU fallback")

    t0 = time.time()
    lp = make_lp(args)
    gen_s = time.time() - t0
    basis = mature_basis(args)

    if args.pmc_child:
        # profiled child: the same warm-up + timed pivots, eager launches (one dispatch per kernel either way)
        eng = make_engine(args, lp, 0, basis)
        eng.set_option("use_graph", 0)
        eng.set_option("check_every", 1)  # no launches beyond the step limit: the last K dispatches are the timed pivots
        eng.set_option("row_price_frac", 0.0)  # the by-column sweep the roofline is quoted for
        if basis is not None and args.preroll > 0:
            eng.dual_steps(args.preroll)
        eng.dual_steps(args.warmup)
        eng.dual_steps(args.steps)
        torch.cuda.synchronize()
        return

    def attach(eng):
        if distributed or os.environ.get("CLPGPU_FORCE_COMM"):
            from clp_amd.multigpu import attach_communicator

            attach_communicator(eng, rank, world)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- headline: warm-up (startup: factorize, resync + W pivots), then exactly K timed pivots
    eng = make_engine(args, lp, local_rank, basis)
    attach(eng)
    preroll = args.preroll if (basis is not None and args.preroll > 0) else 0
    if preroll:
        assert eng.dual_steps(preroll) == -1  # context preparation, not warm-up: see --preroll
    status = eng.dual_steps(args.warmup)
    assert status == -1, f"LP finished during warmup (status {status})"
    it0 = eng.numberIterations()
    nucleus_at_start = int(eng.stats()["nucleus"])
    barrier()
    t1 = time.perf_counter()
    status = eng.dual_steps(args.steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    barrier()
    steps_done = eng.numberIterations() - it0
    assert steps_done == args.steps, f"timed {steps_done} pivots, wanted {args.steps} (status {status})"
    headline_stats = eng.stats()
    if distributed:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- secondary: the same count of pivots from the slack basis (near-identity basis, pricing by row, nucleus <= W + K)
    slack_start = None
    if basis is not None and rank == 0 and world == 1:
        es = make_engine(args, lp, local_rank)
        es.dual_steps(args.warmup)
        torch.cuda.synchronize()
        ts = time.perf_counter()
        es.dual_steps(args.steps)
        torch.cuda.synchronize()
        ts = time.perf_counter() - ts
        slack_start = {"value": args.steps / ts, "unit": "iterations/s", "pivot_window": [args.warmup + 1, args.warmup + args.steps],
                       "nucleus_at_end_of_window": int(es.stats()["nucleus"]),
                       "note": "the same LP from the slack basis: near-identity basis, pi sparse, pricing by row -- what rounds 1-3 quoted as the headline"}
        del es

    # ---- replay legs: the same pivots again on fresh contexts (the engine is deterministic), eager launches + HIP events
    def replay(force_by_column):
        e = make_engine(args, lp, local_rank, basis)
        attach(e)
        e.set_option("timing", 2)
        if force_by_column:
            e.set_option("row_price_frac", 0.0)  # bit-identical tableau rows either way: same pivots
        if preroll:
            e.dual_steps(preroll)
        e.dual_steps(args.warmup)
        torch.cuda.synchronize()
        a0, b0 = e.stats(), e.kernelTimes()
        e.dual_steps(args.steps)
        torch.cuda.synchronize()
        a1, b1 = e.stats(), e.kernelTimes()
        same = bool((e.pivotLog()["sequenceIn"][: it0 + args.steps] == eng.pivotLog()["sequenceIn"][: it0 + args.steps]).all())
        kern = {}
        for name, (ms, cnt) in b1.items():
            ms0, cnt0 = b0.get(name, (0.0, 0))
            if cnt - cnt0 > 0:
                kern[name] = round(1e3 * (ms - ms0) / args.steps, 2)  # per PIVOT (a kernel may launch twice)
        return {k: a1[k] - a0[k] for k in ("price_ms", "price_launches", "price_bytes", "row_ms", "row_launches", "row_bytes")}, kern, same

    # (1) as the headline runs (row pricing by row while pi is sparse): per-kernel times, the by-row form's numbers
    d_mix, per_kernel_us, same_pivots = replay(False)
    # (2) row pricing forced by column: the HBM-bound sweep the roofline is quoted for, on the same pivots -- only when the
    # timed window priced by row somewhere (the slack start); from the mature basis pi is dense and (1) IS the by-column form
    if d_mix["row_launches"] > 0:
        d_col, per_kernel_us_col, same_pivots_col = replay(True)
    else:
        d_col, per_kernel_us_col, same_pivots_col = d_mix, per_kernel_us, same_pivots
    launches = d_col["price_launches"]
    per_launch_bytes = d_col["price_bytes"] / max(launches, 1)
    per_launch_s = d_col["price_ms"] * 1e-3 / max(launches, 1)
    achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
    row_pricing = None
    if d_mix["row_launches"] > 0:
        rb = d_mix["row_bytes"] / d_mix["row_launches"]
        rs = d_mix["row_ms"] * 1e-3 / d_mix["row_launches"]
        row_pricing = {"launches": int(d_mix["row_launches"]), "of_pivots": args.steps, "bytes_per_launch": rb, "us_per_launch": rs * 1e6,
                       "achieved": rb / rs / 1e9 if rs > 0 else 0.0, "unit": "GB/s",
                       "note": "pricing by row (sparse pi): algorithmic bytes B_row = 12 B per visited row entry and pi nonzero + 8 per touched "
                               "column + 20 per emitted one (SURVEY 8d); a scatter bounded by atomics and latency, not by HBM"}
    price_names = [n for n in per_kernel_us_col if n.startswith("k_price")]

    # ---- HBM traffic of the pricing kernel from PMC counters (separate rocprofv3 passes of this script)
    traffic, traffic_source = None, None
    default_workload = (args.workload, args.rows, args.cols, args.nnz_per_col) == ("sparse", 50000, 200000, 50)
    if rank == 0 and world == 1 and args.pmc == "auto":
        traffic, traffic_source = pmc_traffic(args, "k_price")
        if traffic is None:
            why = traffic_source
            for name in ("r02_pmc_price.json", "r01_pmc_price_sell.json"):
                path = os.path.join(ROOT, "profiles", name)
                if os.path.exists(path) and default_workload:
                    traffic = json.load(open(path))["traffic_bytes_per_launch"]
                    traffic_source = f"profiles/{name} (committed rocprofv3 --pmc summary of the default bench line; live pass unavailable: {why})"
                    break
    # ---- the kernel's own duration: a rocprofv3 --kernel-trace pass over the same pivots.  The HIP-event figure above brackets an EAGER
    # launch (event pair + launch gap: ~6 us for an empty kernel of k_price_lds' launch shape); the timed window itself replays hipGraphs,
    # where that gap does not exist.  `roofline.frac` is quoted on the kernel's duration, the event figure stays as `eager_events`.
    ktrace, ktrace_source = None, None
    if rank == 0 and world == 1 and args.pmc == "auto" and price_names:
        main_kernel = "k_price_lds" if "k_price_lds" in price_names else ("k_price_sell" if "k_price_sell" in price_names else "k_price")
        ktrace, ktrace_source = kernel_trace_us(args, main_kernel)
    events = {"us_per_launch": per_launch_s * 1e6, "achieved": achieved, "frac": achieved / HBM_PEAK_GBS,
              "note": "HIP events on the engine's stream around each EAGER pricing launch of the replayed window: kernel + event pair + launch gap"}
    duration_source = "HIP events around eager launches (no rocprofv3 pass: " + str(ktrace_source) + ")"
    if ktrace:
        per_launch_s = ktrace["avg_us"] * 1e-6
        achieved = per_launch_bytes / per_launch_s / 1e9
        duration_source = ktrace_source
    moved = traffic / per_launch_s / 1e9 if (traffic and per_launch_s > 0) else None
    # the committed rocprofv3 summary of this command (profiles/, tools/closing_bench.sh): the kernel's own duration, without the
    # event pair and the eager launch gap the live figure above includes (an empty kernel of k_price_lds' launch shape costs 6 us there)
    profile_ref = None
    if default_workload:
        for name in ("r05_bench_kernel_stats.txt",):
            path = os.path.join(ROOT, "profiles", name)
            if os.path.exists(path):
                for line in open(path):
                    m_ = re.match(r"\s*clpgpu::(k_price_lds|k_price_sell)\(.*?\)\s+(\d+)\s+(\d+)\s+([0-9.]+)", line)
                    if m_ and m_.group(1) in " ".join(price_names):
                        us = float(m_.group(4))
                        profile_ref = {"file": f"profiles/{name}", "kernel": m_.group(1), "launches": int(m_.group(2)), "avg_us": us,
                                       "frac_on_the_live_bytes": (per_launch_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS) if us > 0 else None,
                                       "note": "rocprofv3 --kernel-trace --stats of `python bench.py --cpu-iterations 0 --pmc off --tto-budget 0 --ladder-budget 0` on the MI355X, committed"}
                        break

    # ---- CPU baseline: the oracle over >= 500 pivots and >= 2 s of the same LP, and the engine over the same pivots
    cpu = same_window = clp = None
    if rank == 0 and world == 1 and args.cpu_iterations != 0:
        from oracle.oracle import OracleSimplex

        n_cpu = args.cpu_iterations if args.cpu_iterations > 0 else 500
        while True:
            o = OracleSimplex(lp)
            o.set_option("pivot_rule", args.pivot_rule)
            o.set_option("max_pivots", 0)
            o.set_option("max_iterations", n_cpu)
            o.dual(live=True)  # timed HERE: never answered from a committed record
            if args.cpu_iterations > 0 or o.seconds >= 5.0 or o.iterations < n_cpu or n_cpu >= 2000:
                break
            n_cpu += 500  # 500 -> 1000 -> 1500 -> 2000 pivots: 0.4 / 1.4 / 3.3 / ~12 s of one Xeon core at config 4 (dense nucleus LU: k^3; 2500: 42 s)
        cpu = {"value": o.iterations / max(o.seconds, 1e-9), "unit": "iterations/s", "cores": 1, "kind": "port",
               "window": [1, int(o.iterations)], "seconds": round(o.seconds, 3),
               "sample": f"pivots 1..{o.iterations} of the same LP from the slack basis ({o.seconds:.2f} s, refactorizations included, "
                         "MPS/generation excluded), CPU oracle = C restatement of ClpSimplexDual with a dense nucleus LU "
                         "(the reference needs CoinUtils and cannot be built here; it is NOT Clp)"}
        eng_c = make_engine(args, lp, local_rank)
        eng_c.dual_steps(0)  # startup only
        torch.cuda.synchronize()
        tc = time.perf_counter()
        eng_c.dual_steps(int(o.iterations))
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        same_window = {"value": o.iterations / tc, "unit": "iterations/s", "window": [1, int(o.iterations)], "seconds": round(tc, 4),
                       "speedup_vs_cpu_port": (o.iterations / tc) / cpu["value"],
                       "same_pivots_as_cpu": bool((eng_c.pivotLog()["sequenceIn"][: o.iterations] == o.pivot_log()["sequenceIn"]).all())}
        cpu["gpu_same_window"] = same_window
        del eng_c
        # ---- the headline's OWN regime, live on this box: the oracle warm-started from the committed mature basis for a bounded number of
        # pivots.  Its start-up (a dense LU of the order-10 514 nucleus: ~15 s with the blocked, threaded elimination; twenty minutes with the
        # unblocked loop of rounds 1-5) is clocked apart (orc_startup_seconds) and left out on BOTH sides: `value` is pivots per second of
        # one core once the basis is factorized, next to the engine over exactly those pivots after ITS start-up factorization.
        if basis is not None and len(basis) == lp.m + lp.n and args.cpu_mature_pivots > 0:
            om = OracleSimplex(lp)
            om.set_option("pivot_rule", args.pivot_rule)
            om.set_option("max_pivots", 0)
            # both sides under ClpDualRowSteepest's full scan in this leg: the default mode 3 scans the first numberWanted rows of the
            # infeasibility list in BASIS-POSITION order, and two factorizations (the oracle's dense LU, the engine's front + tail) put
            # the basic variables at different positions -- with it the two sides would time different pivots
            om.set_option("steepest_mode", 1)
            om.set_status((basis & 7).astype(np.uint8))
            om.set_option("max_iterations", args.cpu_mature_pivots)
            om.dual(live=True)
            n_m = int(om.iterations)
            piv_s = max(om.seconds - om.startup_seconds, 1e-9)
            em = make_engine(args, lp, local_rank, (basis & 7).astype(np.uint8))
            em.set_option("steepest_mode", 1)
            em.dual_steps(0)  # start-up factorization of the basis: before the clock, as on the CPU side
            torch.cuda.synchronize()
            tm = time.perf_counter()
            em.dual_steps(n_m)
            torch.cuda.synchronize()
            tm = time.perf_counter() - tm
            lo, lg = om.pivot_log(), em.pivotLog()
            n_same = 0
            while n_same < min(len(lo), len(lg)) and lo[n_same]["sequenceIn"] == lg[n_same]["sequenceIn"] and lo[n_same]["sequenceOut"] == lg[n_same]["sequenceOut"]:
                n_same += 1
            slack_pair = cpu
            cpu = {"value": n_m / piv_s, "unit": "iterations/s", "cores": 1, "kind": "port",
                   "regime": "the headline's: config 4 warm-started from the committed mature basis (nucleus 10 514, pi dense)",
                   "window": [1, n_m], "seconds": round(piv_s, 3),
                   "startup_seconds_excluded": round(om.startup_seconds, 2),
                   "startup_threads": int(os.environ.get("OMP_NUM_THREADS", os.cpu_count() or 1)),
                   "sample": f"pivots 1..{n_m} from the committed mature basis of the same LP, timed on this box: {piv_s:.2f} s of one core for the pivots "
                             f"(dense LU solves of order 10 514: 4 per pivot); the start-up factorization ({om.startup_seconds:.1f} s, the only threaded "
                             "part) is excluded here and on the GPU side.  CPU oracle = C restatement of ClpSimplexDual with a dense nucleus LU "
                             "(the reference needs CoinUtils and cannot be built here; it is NOT Clp)",
                   "gpu_same_window": {"value": n_m / tm, "unit": "iterations/s", "window": [1, n_m], "seconds": round(tm, 4),
                                       "excludes": "the start-up factorization (clpgpu_dual_steps(0) before the clock)",
                                       "speedup_vs_cpu_port": (n_m / tm) / (n_m / piv_s), "identical_pivots": int(n_same), "of": n_m,
                                       "steepest_mode": "1 on both sides in this leg (full CHUZR scan: position-independent, so that both time the same pivots)"},
                   "slack_start": slack_pair}
            del em, om
        clp = clp_upstream(args, lp)
        cpu["clp_upstream"] = clp

    # ---- time to optimal + sustained rate: the headline context carries on until optimal or the budget runs
    # out, in chunks of 2000 pivots; every chunk records pivots/s, the nucleus and the LU split, so the line says
    # what the engine sustains once the basis has matured -- not only what it does on the near-identity basis of
    # the timed window
    tto = sustained = regime = None
    optimum = None
    if rank == 0 and world == 1 and args.tto_budget > 0:
        t2 = time.perf_counter()
        st = -1
        chunks = []
        chunk = 2000
        while st == -1 and time.perf_counter() - t2 < args.tto_budget:
            i0, c0 = eng.numberIterations(), time.perf_counter()
            st = eng.dual_steps(chunk)
            torch.cuda.synchronize()
            c1 = time.perf_counter()
            info = eng.stats()
            chunks.append({"pivots": [int(i0) + 1, int(eng.numberIterations())], "seconds": round(c1 - c0, 4),
                           "iterations_per_s": round((eng.numberIterations() - i0) / max(c1 - c0, 1e-9), 1),
                           "nucleus": int(info["nucleus"]), "lu": bool(info["lu_active"]), "lu_front": int(info["lu_front"]),
                           "lu_tail": int(info["lu_tail"]), "objective": eng.objectiveValue()})
        spent = time.perf_counter() - t2
        # startup + warm-up + timed window ran before t2: total_ms of the engine covers them all
        total_s = eng.stats()["total_ms"] * 1e-3
        info = eng.stats()
        opt_path = os.path.join(ROOT, "tests", "golden", "bench_optima.json")
        if os.path.exists(opt_path):
            optimum = json.load(open(opt_path)).get(lp.name)
        tto = {"status": int(st), "iterations": int(eng.numberIterations()), "engine_seconds": round(total_s, 3),
               "time_to_optimal_s": round(total_s, 3) if st == 0 else None, "objective": eng.objectiveValue(),
               "nucleus": int(info["nucleus"]), "refactorizations": int(info["refactorizations"]),
               "independent_optimum": optimum,
               "objective_matches_independent": (abs(eng.objectiveValue() - optimum["objective"]) <= 1e-8 * abs(optimum["objective"]))
               if (st == 0 and optimum and optimum.get("objective") is not None) else None,
               "note": "optimal" if st == 0 else f"did not finish within the {args.tto_budget:.0f} s budget (continued {spent:.1f} s past the timed window); "
                       "the dual objective is a lower bound that rises monotonically, see DESIGN section 6.3 for why this LP is out of reach"}
        if chunks:
            pick = [chunks[0]]
            mid = next((c for c in chunks if c["nucleus"] >= 5000), None)
            if mid and mid is not chunks[0]:
                pick.append(mid)
            if chunks[-1] is not pick[-1]:
                pick.append(chunks[-1])
            whole = (chunks[-1]["pivots"][1] - chunks[0]["pivots"][0] + 1) / max(sum(c["seconds"] for c in chunks), 1e-9)
            objs = [c["objective"] for c in chunks]
            sustained = {"unit": "iterations/s", "windows": pick, "over_the_whole_leg": round(whole, 1),
                         # the dual objective is a lower bound on the optimum and must not fall (cost shifting aside): checked per window
                         "dual_objective_monotone": bool(all(b >= a - 1e-9 * abs(a) for a, b in zip(objs, objs[1:]))),
                         "dual_objective_first_last": [objs[0], objs[-1]],
                         # what the solve gains, next to how fast it pivots: on this LP the two do not go together
                         "dual_objective_gain_per_s": (round((objs[-1] - objs[0]) / max(sum(c["seconds"] for c in chunks[1:]), 1e-9), 3)
                                                       if len(chunks) > 1 else None),
                         "mature": pick[-1]["iterations_per_s"],
                         "note": "wall clock around clpgpu_dual_steps(2000) on the live solve, refactorizations (host Markowitz front + dense tail "
                                 "inversion) included; `mature` = the last window reached within the budget"}
        # ---- per-kernel times and rooflines of the MATURE regime: 256 more pivots with eager launches and a HIP event after
        # every launch (option timing 2), on the same live context
        if st == -1:
            i0 = eng.stats()
            eng.set_option("timing", 2)
            k0 = eng.kernelTimes()
            eng.dual_steps(256)
            torch.cuda.synchronize()
            k1, i1 = eng.kernelTimes(), eng.stats()
            n_piv = 256
            kern = {}
            for name, (ms, cnt) in k1.items():
                ms0, cnt0 = k0.get(name, (0.0, 0))
                if cnt - cnt0 > 0:
                    kern[name] = (1e3 * (ms - ms0) / (cnt - cnt0), cnt - cnt0)  # us per launch, launches
            lu = bool(i1["lu_active"])
            kd = float(i1["lu_tail"] if lu else i1["nucleus"])  # order of the dense matrix a solve streams
            eta = 0.5 * (i0["eta_count"] + i1["eta_count"]) if i1["eta_count"] >= i0["eta_count"] else 0.5 * i1["eta_count"]
            # (the launch counters switch source with the timing option; the byte counters are the device's own throughout)
            by_row = int(round((i1["row_bytes"] - i0["row_bytes"]) > 0))
            n_price = kern.get("k_price_lds", kern.get("k_price_sell", (0.0, n_piv)))[1]
            price_b = (i1["price_bytes"] - i0["price_bytes"]) / max(n_price, 1) if i1["price_bytes"] > i0["price_bytes"] else None
            model = {  # algorithmic bytes per launch of the kernels that stream a matrix in this regime
                "k_gemv3g": (8.0 * kd * kd, "hbm", "three FTRAN right-hand sides through the dense " + ("tail inverse" if lu else "nucleus inverse") + ": 8 k^2 B"),
                "k_lu_gemvT": (8.0 * kd * kd, "hbm", "BTRAN through the transposed tail inverse: 8 k^2 B"),
                "k_lu_gemv3": (8.0 * kd * kd, "hbm", "three FTRAN right-hand sides through the dense tail inverse in one sweep: 8 k^2 B"),
                "k_ftran_scatter3_lu": ((8.0 * lp.m * eta + 3 * 8.0 * lp.m, "hbm", "eta file: x = x0 - H s over the m positions, 8 m t B (t = etas since the factorization)")
                                        if not i1["eta_compact_slots"] else
                                        (None, "latency", "compact eta file: the scatter, DSE weights and the slack positions from their own rows (gathers)")),
                "k_lu_eta_apply": (8.0 * float(i1["eta_compact_slots"]) * eta, "hbm",
                                   "compact eta file: x_K -= Hc s over the positions that hold or held a structural, 8 (k + conversions) t B instead of 8 m t"),
                "k_primal_rank1": (16.0 * kd * kd, "hbm", "rank-1 update of the explicit nucleus inverse: 16 k^2 B"),
                "k_price_sell": (None if "k_price_lds" in kern else price_b, "hbm", "row pricing by column: bytes the kernel streams (4 B per row index, 8 B per element fetched, lists)"),
                "k_price_lds": (price_b, "hbm", "row pricing by column with pi tiles in LDS (jagged tile-by-tile streams of the SELL windows, fused first ratio pass): SURVEY 8d's B_col = 12 B per entry of the scanned columns + lists"),
            }
            rl = []
            for name, (us, cnt) in sorted(kern.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                ent = {"kernel": name, "us_per_launch": round(us, 2), "launches_per_pivot": round(cnt / n_piv, 2),
                       "share_of_kernel_time": round(us * cnt / max(sum(u * c for u, c in kern.values()), 1e-9), 3)}
                if name in model and model[name][0]:
                    by = model[name][0]
                    ent.update({"bound": model[name][1], "bytes_per_launch": by, "achieved": by / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, "what": model[name][2]})
                else:
                    ent["bound"] = "latency"
                rl.append(ent)
            regime = {"pivots": [int(i0["iterations"]) + 1, int(i1["iterations"])], "nucleus": int(i1["nucleus"]), "lu": lu,
                      "lu_front": int(i1["lu_front"]), "lu_tail": int(i1["lu_tail"]), "eta_file_mean_length": eta,
                      "pricing": "by row on some pivots" if by_row else "by column on every pivot (pi is dense)",
                      "kernels": rl[:14],
                      "note": "eager launches with a HIP event after each (kernel + launch gap); `frac` = algorithmic bytes / time / 8 TB/s"}

    # ---- time-to-optimal ladder: the same generator at sizes HiGHS finishes; the engine in its default mode, slack start, to status 0
    ladder = None
    if rank == 0 and world == 1 and args.ladder_budget > 0:
        from tools.ladder import ladder_lp

        gold_path = os.path.join(ROOT, "tests", "golden", "ladder_optima.json")
        gold = json.load(open(gold_path)) if os.path.exists(gold_path) else {}
        ladder = {"unit": "seconds to status 0 from the slack basis, default options (steepest edge, LU mode from 3072 basic structurals on)",
                  "independent_solver": "HiGHS serial dual simplex (scipy), presolve off, one core -- tests/golden/ladder_optima.json, tools/ladder.py",
                  "rungs": []}
        t_l = time.perf_counter()
        for name in filter(None, args.ladder_rungs.split(",")):
            ref = gold.get(name, {}).get("highs")
            kkt_ref = gold.get(name, {}).get("kkt")
            if not ref or ref.get("objective") is None:
                ref = None
                if not kkt_ref or kkt_ref.get("objective") is None:
                    continue
            if time.perf_counter() - t_l > args.ladder_budget:
                ladder["rungs"].append({"rung": name, "skipped": "ladder budget spent"})
                continue
            llp = ladder_lp(name)
            el = make_engine(args, llp, local_rank)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            stl = -1
            while stl == -1 and time.perf_counter() - t_l < args.ladder_budget + 30.0:
                stl = el.dual_steps(4000)
            torch.cuda.synchronize()
            t3 = time.perf_counter() - t3
            obj = el.objectiveValue()
            orc = gold.get(name, {}).get("oracle")
            if ref is None:
                # no independent solver finishes this rung: optimality from a certificate computed outside the engine, here and now
                from tools.kkt_certificate import certify, row_duals_from_engine

                cert = certify(llp, el.solution(), row_duals_from_engine(llp, el)) if stl == 0 else None
                ladder["rungs"].append({
                    "rung": name, "rows": int(llp.m), "cols": int(llp.n), "nnz": int(len(llp.elem)), "status": int(stl),
                    "engine_seconds": round(t3, 3) if stl == 0 else None, "engine_iterations": int(el.numberIterations()),
                    "engine_iterations_per_s": round(el.numberIterations() / max(t3, 1e-9), 1), "objective": obj,
                    "independent_check": "KKT certificate computed outside the engine (tools/kkt_certificate.py): HiGHS hit its ten-hour limit on this rung",
                    "certificate": cert, "certified_optimal": bool(cert and cert["optimal"]) if stl == 0 else None,
                    "note": None if stl == 0 else "not finished within the ladder budget (100-270 s on the MI355X depending on the trajectory: python tools/ladder.py engine 7000)",
                    "committed_objective": kkt_ref["objective"],
                    "objective_matches_committed": bool(stl == 0 and abs(obj - kkt_ref["objective"]) <= 1e-8 * abs(kkt_ref["objective"])),
                    "nucleus_at_end": int(el.stats()["nucleus"])})
                del el
                continue
            ladder["rungs"].append({
                "rung": name, "rows": int(llp.m), "cols": int(llp.n), "nnz": int(len(llp.elem)), "status": int(stl),
                "engine_seconds": round(t3, 3) if stl == 0 else None, "engine_iterations": int(el.numberIterations()),
                "engine_iterations_per_s": round(el.numberIterations() / max(t3, 1e-9), 1), "objective": obj,
                "highs_objective": ref["objective"], "highs_seconds": ref["seconds"], "highs_iterations": ref["iterations"],
                "objective_matches_highs": bool(stl == 0 and abs(obj - ref["objective"]) <= 1e-8 * abs(ref["objective"])),
                "speedup_vs_highs": round(ref["seconds"] / t3, 2) if stl == 0 else None,
                "oracle_port_seconds": orc["seconds"] if orc and orc.get("status") == 0 else None,
                "oracle_port_iterations": orc["iterations"] if orc and orc.get("status") == 0 else None,
                "nucleus_at_end": int(el.stats()["nucleus"])})
            del el

    # ---- what the ladder says about config 4's distance to its optimum: pivots to optimality against rows over the finished rungs
    # (same generator, same density rule), least squares in log-log, extrapolated to this LP's rows
    if tto is not None and tto.get("time_to_optimal_s") is None and ladder and default_workload:
        done = [(r["rows"], r["engine_iterations"]) for r in ladder["rungs"] if r.get("status") == 0]
        if len(done) >= 3:
            lx, ly = np.log([d[0] for d in done]), np.log([d[1] for d in done])
            slope, icpt = np.polyfit(lx, ly, 1)
            need = float(np.exp(icpt + slope * np.log(lp.m)))
            rate = (sustained or {}).get("mature") or (args.steps / elapsed)
            tto["extrapolation"] = {
                "law": f"pivots to optimality ~ rows^{slope:.2f} over the rungs finished in this run ({', '.join(str(d[0]) for d in done)} rows: "
                       f"{', '.join(str(d[1]) for d in done)} pivots)",
                "pivots_at_this_size": round(need), "at_the_sustained_rate_s": round(need / max(rate, 1e-9)),
                "note": "an extrapolation of the same generator's smaller instances, not a measurement -- and too low: a 45-minute run of this LP from the "
                        "slack basis made 3.74 M pivots (1 382 it/s sustained, dual objective 768 399) without reaching the optimum, "
                        "profiles/r06_config4_long.jsonl; the solve above is "
                        f"{tto['iterations']} pivots in (counted from the committed basis at pivot 30 000 when started there)"}

    # ---- BASELINE configs[2] and the Netlib-shaped variant as sub-records (children of this script on the same GPU, one after the other)
    sub_records = None
    if rank == 0 and world == 1 and default_workload and args.sub_records != "off":
        sub_records = {}
        shapes = {"dense": ["--rows", "5000", "--cols", "5000", "--steps", "300", "--warmup", "100", "--tto-budget", "30"],
                  "netlib": ["--rows", "50000", "--cols", "200000", "--steps", "500", "--warmup", "200", "--tto-budget", "15"]}
        for w in filter(None, args.sub_records.split(",")):
            if w not in shapes:
                continue
            cmd = [sys.executable, os.path.abspath(__file__), "--workload", w, *shapes[w], "--cpu-iterations", "0", "--pmc", "off",
                   "--ladder-budget", "0", "--sub-records", "off", "--shard-proxy", "off", "--start", "slack"]
            t_sub = time.perf_counter()
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=args.sub_timeout)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if r.returncode != 0 or not line:
                    sub_records[w] = {"error": f"child rc {r.returncode}", "tail": (r.stderr or r.stdout)[-300:]}
                    continue
                c = json.loads(line[-1])
                rf = c["roofline"]
                sub_records[w] = {
                    "workload": c["config"]["workload"], "value": c["value"], "unit": c["unit"], "ms_per_step": c["ms_per_step"],
                    "pivot_window": c["config"]["pivot_window"], "window_counts_from": "the slack basis",
                    "pricing": {"kernel": rf["kernel"], "bytes_per_launch": rf["bytes_per_launch"], "us_per_launch": rf["us_per_launch"],
                                "achieved": rf["achieved"], "frac": rf["frac"], "duration_source": rf["duration_source"], "form_in_timed_window": rf["form_in_timed_window"]},
                    "time_to_optimal": c["time_to_optimal"], "sustained": (c["sustained"] or {}).get("over_the_whole_leg"),
                    "refactorizations": c["refactorizations"], "child_seconds": round(time.perf_counter() - t_sub, 1)}
                prof = os.path.join(ROOT, "profiles", f"r06_mfma_{w}.txt")
                if os.path.exists(prof):
                    sub_records[w]["mfma_utilisation_profile"] = {"file": f"profiles/r06_mfma_{w}.txt", "text": open(prof).read()[-600:]}
            except subprocess.TimeoutExpired:
                sub_records[w] = {"error": f"child did not finish in {args.sub_timeout:.0f} s"}

    # ---- what a rank of an R-GPU run launches per pivot, timed on this one GPU: the pricing kernel over the column shard [0, n / R)
    # with a dense pi (the mature regime), for each R.  No 8-GPU node exists for this repository: this is the single-GPU proxy of the
    # pricing part of the scaling curve; the two small list exchanges per pivot are NOT in it (loopback copies would
    # time a host barrier, not xGMI).
    shard_proxy = None
    if rank == 0 and world == 1 and default_workload and args.shard_proxy != "off":
        shard_proxy = {"unit": "us per launch (HIP events, 30 launches, dense pi)", "bytes": "SURVEY 8d B_col of the shard's columns", "ranks": []}
        col_len = np.diff(np.asarray(lp.col_start))
        for R in (int(x) for x in args.shard_proxy.split(",")):
            last = (lp.n // R + 255) // 256 * 256 if R > 1 else lp.n
            last = min(last, lp.n)
            nnz_shard = int(col_len[:last].sum())
            b_col = 12.0 * nnz_shard + 4.0 * (last + 1) + last + 8.0 * lp.m + 20.0 * last
            ent = {"ranks": R, "columns": int(last), "nnz": nnz_shard, "bytes_per_launch": b_col}
            try:
                from clp_amd.engine import ClpGpuSimplex

                e = ClpGpuSimplex(local_rank)
                e.set_option("price_lds_min_windows", 1)  # before the load: lays the shard out for the LDS form too, however narrow
                e.loadProblem(lp)
                e.set_option("pivot_rule", args.pivot_rule)
                e.set_option("max_pivots", 0)
                if R > 1:
                    e.setColumnRange(0, last)
                e.dual_steps(0)
                for form, us in zip(("k_price_sell", "k_price_lds"), e.debugPriceBench([0, 1 << 20], reps=30)):
                    us = float(us)
                    if us > 0:
                        ent[form] = {"us": round(us, 2), "achieved": b_col / (us * 1e-6) / 1e9, "frac": b_col / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                     "aggregate_over_ranks": R * b_col / (us * 1e-6) / 1e9}
                del e
            except Exception as ex:  # noqa: BLE001 -- a probe must not take the line down
                ent["error"] = str(ex)[:200]
            shard_proxy["ranks"].append(ent)

    config_ref = {"sparse": "BASELINE.json configs[3]", "dense": "BASELINE.json configs[2]",
                  "netlib": "Netlib-shaped variant of BASELINE.json configs[3]"}[args.workload]
    if (args.rows, args.cols) != {"dense": (5000, 5000)}.get(args.workload, (50000, 200000)):
        config_ref += ", non-default size"
    # the kernel that takes the most time per pivot is not the one that moves the bytes: say which, and what bounds it
    kernel_bound = {
        "k_dual_column": ("latency", "bound-flipping ratio test: one workgroup, candidates in registers (28 B each), about 11 dependent "
                                     "wave-wide reductions per pivot -- no byte or flop roofline applies (DESIGN section 5)"),
        "k_price_sell": ("hbm", "row pricing; by column this is the HBM sweep the roofline object is quoted for"),
        "k_ftran_scatter3": ("latency", "scatter of three FTRAN results + DSE weights, 16-40 B x rows"),
        "k_fix_house": ("latency", "inverse fix-up + housekeeping, serial tail"),
        "k_chuzr_scan": ("latency", "CHUZR scan over the infeasibility list"),
        "k_primal_rank1": ("hbm", "rank-1 update of the nucleus inverse: 16 k^2 bytes"),
        "k_gemv3g": ("hbm", "three FTRAN right-hand sides through the nucleus inverse: 8 k^2 bytes"),
    }
    dominant_by_time = None
    if per_kernel_us:
        top = max(per_kernel_us, key=per_kernel_us.get)
        dominant_by_time = {"kernel": top, "us_per_pivot": per_kernel_us[top],
                            "share_of_kernel_time": round(per_kernel_us[top] / max(sum(per_kernel_us.values()), 1e-9), 3),
                            "bound": kernel_bound.get(top, ("latency", ""))[0], "note": kernel_bound.get(top, ("latency", "latency-bound small kernel"))[1]}
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
                                   + ("steepest-edge dual" if args.pivot_rule else "Dantzig dual")
                                   + (" warm-started from the committed basis of this LP after 30 000 pivots (tests/golden/basis_sparse_30000.npy)"
                                      if basis is not None else " from the slack basis"),
                       "start": "mature basis (pivot 30 000 of the same LP)" if basis is not None else "slack basis",
                       "preparation": (f"start-up factorization of that basis + {preroll} untimed pivots, so that the window sits mid-way through an "
                                       "eta-file cycle; then the W warm-up pivots") if basis is not None else None,
                       "rows": int(lp.m), "cols": int(lp.n), "nnz": int(len(lp.elem)),
                       "parallelism": f"column-range pricing x{world}" if world > 1 else "1 GPU",
                       "check_every": args.check_every, "generate_s": round(gen_s, 1),
                       "pivot_window": [int(it0) + 1, int(it0) + args.steps],
                       "pivot_window_counts_from": "the warm start" if basis is not None else "the slack basis",
                       "nucleus_at_start_of_window": nucleus_at_start, "nucleus_at_end_of_window": int(headline_stats["nucleus"])},
            "roofline": {"bound": "hbm",
                         # by column the HIP events bracket the pricing kernel alone (k_price_row_finish returns at once and is left out): the
                         # duration rocprofv3 reports for k_price_sell; by row both passes
                         "kernel": ("k_price_lds" if "k_price_lds" in price_names else
                                    ("k_price_sell" if (d_col is d_mix and "k_price_sell" in price_names) else " + ".join(sorted(price_names))))
                                   + " (row pricing + fused first ratio pass)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "bytes_per_launch": per_launch_bytes, "us_per_launch": per_launch_s * 1e6,
                         "duration_source": duration_source, "kernel_trace": ktrace, "eager_events": events,
                         "launches": int(launches),
                         "window": ("replay of the timed pivots (eager, HIP events on the engine's stream): pi is dense in this regime, so the "
                                    "kernel form measured here (by column) is the one the timed window ran"
                                    if d_col is d_mix else
                                    "FORCED-BY-COLUMN REPLAY of the timed pivots (eager, HIP events): the HBM-bound form of the pricing kernel. "
                                    "The timed window itself prices by row while pi is sparse -- `row_pricing` is what ran there, "
                                    "`roofline_mature` what runs once the basis has matured (pi dense, pricing by column)"),
                         "form_in_timed_window": ("by row" if (row_pricing and row_pricing["launches"] * 2 > args.steps) else "by column"),
                         "bytes_counted": ("SURVEY 8d's B_col = 12 nnz(A_J) + 4 |J| + n + 8 m + 20 nnz_out, counted by the kernel (k_price_lds itself streams "
                                           "20-byte records of two entries: 10 B per entry + 3 % pair padding)") if "k_price_lds" in price_names else
                                          ("streamed: 4 B per row index + 8 B per element fetched (conditional form fetches only under set bits of pi) "
                                           "+ lists; SURVEY 8d's B_col = 12 nnz(A_J) + ... is the unconditional form"),
                         "replay_identical": bool(same_pivots and same_pivots_col),
                         "row_pricing": row_pricing,
                         "rocprofv3_reference": profile_ref,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "moved": moved, "moved_frac": (moved / HBM_PEAK_GBS) if moved else None,
                         "per_kernel_us": per_kernel_us,
                         "dominant_by_time": dominant_by_time,
                         "per_kernel_note": "per pivot, eager launches: kernel + the launch gap before it; hipGraph replay (the headline) has smaller gaps"},
            "cpu_baseline": cpu,
            "slack_start": slack_start,
            "time_to_optimal": tto,
            "time_to_optimal_ladder": ladder,
            "sustained": sustained,
            "roofline_mature": regime,
            "sub_records": sub_records,
            "shard_pricing_proxy": shard_proxy,
            "refactorizations": int(headline_stats["refactorizations"]),
            # what sent the iteration loop to its status checks up to the end of the timed window (src/ClpSimplexDual.cpp:1849 scheduled,
            # :1451 alpha check, :1574 objective going backwards, :1618 bad update), and the chain's pricing form
            "status_check_causes": {k: int(headline_stats[k]) for k in ("exits_scheduled", "exits_alpha_check", "exits_backwards", "exits_bad_update")},
            "pricing_form": {"lds_chain_now": int(headline_stats["price_form"]), "dense_pi_launches": int(headline_stats["dense_pi_launches"]),
                             "switches": int(headline_stats["price_form_switches"])},
        }
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()
    # librccl prints a version banner to stdout when the process unloads it (after the line above, for any run that attached a
    # communicator): the bench line must stay the last thing on stdout
    sys.stdout.flush()
    os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
