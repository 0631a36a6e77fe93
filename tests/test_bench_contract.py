"""The bench line's contract (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling / vs_baseline / dtype /
data / config.workload + the `roofline` and `cpu_baseline` objects), checked on the CPU: bench.main() is run against a stand-in engine (no device
here), so this holds the control flow and the JSON shape -- mature start with pre-roll, slack start, the ladder leg -- not any number."""
import collections
import io
import json
import sys
from contextlib import redirect_stdout

import numpy as np
import pytest


class StandIn:
    def __init__(self, device=0):
        self.it, self.warm = 0, False

    def loadProblem(self, lp):
        self.lp = lp
        return self

    def setStatusArray(self, status):
        self.warm = True

    def set_option(self, name, value):
        pass

    def dual_steps(self, count):
        self.it += int(count)
        return 0 if self.it >= 9000 else -1

    def numberIterations(self):
        return self.it

    def objectiveValue(self):
        return 1.0 + self.it

    def stats(self):
        d = collections.defaultdict(float)
        d.update(iterations=self.it, nucleus=7, price_launches=self.it, price_ms=0.05 * self.it, price_bytes=1.0e8 * self.it, total_ms=5.0)
        if not self.warm:
            d.update(row_launches=0.5 * self.it, row_ms=0.01 * self.it, row_bytes=1.0e3 * self.it)
        return d

    def kernelTimes(self):
        return {"k_price_sell": (0.05 * self.it, self.it), "k_dual_column": (0.04 * self.it, self.it)}

    def pivotLog(self):
        from clp_amd.engine import PIVOT_DTYPE

        return np.zeros(self.it + 10, dtype=PIVOT_DTYPE)


@pytest.mark.parametrize("mature", [True, False])
def test_bench_line_shape(monkeypatch, mature):
    import torch

    import bench
    import clp_amd.engine as E

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    monkeypatch.setattr(E, "ClpGpuSimplex", StandIn)
    if mature:
        monkeypatch.setattr(bench, "mature_basis", lambda args: np.zeros(10, dtype=np.uint8))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--rows", "200", "--cols", "800", "--nnz-per-col", "6", "--pmc", "off", "--steps", "40", "--warmup", "8",
                                      "--tto-budget", "0.2", "--ladder-budget", "0", "--cpu-iterations", "60", "--preroll", "30"])
    monkeypatch.setattr(bench.os, "dup2", lambda *a, **k: None)  # (main() parks stdout on /dev/null at the end)
    out = io.StringIO()
    with redirect_stdout(out):
        bench.main()
    lines = [l for l in out.getvalue().splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 8 and d["dtype"] == "f64" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert isinstance(d["config"]["workload"], str) and "model" not in d["config"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert d["roofline"]["bound"] == "hbm" and d["roofline"]["peak"] == 8000.0
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] == 1
    first = 30 + 8 + 1 if mature else 8 + 1
    assert d["config"]["pivot_window"] == [first, first + 39]
    assert (d["slack_start"] is not None) == mature


def test_gpus_without_a_launcher_is_refused(monkeypatch):
    """`python bench.py --gpus 8` with no torch.distributed launcher behind it used to run on one GPU and print n_gpus 1: it must
    stop with the launch line instead (one process per GPU; WORLD_SIZE is what the launcher provides)."""
    import bench

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "torch.distributed.run" in str(e.value) and "--nproc-per-node 8" in str(e.value)
