#!/usr/bin/env python3
"""Writes tests/golden/oracle_cache/: the CPU oracle's long solves that the GPU parity tests compare against, so that a GPU lease is not
spent waiting for one CPU core (oracle/oracle.py, "recorded solves").  Runs HERE, without a GPU: the -m gpu tests are executed with a
stand-in engine class that answers every engine call from the oracle itself, so each test makes exactly the oracle calls it makes on the
GPU box -- same LP, same options, same starting statuses, hence the same record key -- and every solve longer than 2 s is recorded.
(The stand-in proves nothing about the engine; tests that need more than it offers simply fail here, after their oracle calls.)

    python tests/golden/make_oracle_cache.py [pytest selection ...]        # default: every -m gpu test
Records are keyed by the oracle's source hash: after a change to oracle/clp_dual_oracle.c, re-run this script (stale records are
never read; this script deletes them).
    python tests/golden/make_oracle_cache.py --out DIR [pytest selection ...]      # records into DIR (none are read from the committed ones)
    python tests/golden/make_oracle_cache.py --rekey-from OLDHASH [pytest selection ...]
carries the records written under the source hash OLDHASH over to the current one instead of solving again -- for a source change that
cannot touch them (oracle/oracle.py, _REKEY_FROM: only records of unscaled solves, for a change confined to the scaled path)."""
import collections
import os
import sys

if len(sys.argv) > 2 and sys.argv[1] == "--rekey-from":
    os.environ["CLP_ORACLE_REKEY_FROM"] = sys.argv[2]
    del sys.argv[1:3]
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
OUT = os.path.join(HERE, "oracle_cache")
if len(sys.argv) > 2 and sys.argv[1] == "--out":  # write (and look) somewhere else: re-solving committed records to check them
    OUT = os.path.abspath(sys.argv[2])
    del sys.argv[1:3]
os.environ["CLP_ORACLE_CACHE_WRITE"] = OUT
os.environ["CLP_ORACLE_CACHE"] = OUT

import numpy as np  # noqa: E402
import pytest  # noqa: E402
import torch  # noqa: E402

import clp_amd.engine as E  # noqa: E402
from oracle import oracle as O  # noqa: E402


class StandIn:
    """ClpGpuSimplex's surface, answered by the oracle (enough of it for the tests' oracle calls to be reached)"""

    def __init__(self, device=0):
        self.o = None
        self.pending = []
        self.n_done = 0

    def loadProblem(self, lp):
        self.lp, self.m, self.n = lp, int(lp.m), int(lp.n)
        self.o = O.OracleSimplex(lp)
        for k, v in self.pending:
            self.set_option(k, v)
        return self

    def set_option(self, name, value):
        if self.o is None:
            self.pending.append((name, value))
            return
        try:
            self.o.set_option(name, value)
        except KeyError:
            pass

    def setMaximumIterations(self, v):
        self.set_option("max_iterations", v)

    def setStatusArray(self, status):
        self.o.set_status(np.asarray(status, dtype=np.uint8) & 7)

    def _guard(self, pivots):
        # engine-only tests on the full-size LPs (9000 pivots at 50 000 rows ...) would cost the dense-LU oracle hours and are
        # compared with no oracle solve on the GPU box: refuse them here
        if self.m >= 20000 and (pivots is None or self.n_done + pivots > 2600):
            raise RuntimeError("stand-in engine: not an oracle-checked workload")

    def dual(self):
        self._guard(None)
        return self.o.dual()

    def dual_steps(self, count):
        self._guard(int(count))
        self.n_done += int(count)
        self.o.set_option("max_iterations", self.n_done)
        st = self.o.dual()
        return -1 if st == 3 else st

    def numberIterations(self):
        return self.o.iterations

    def objectiveValue(self):
        return self.o.objective

    def problemStatus(self):
        return 0

    def solution(self):
        return self.o.solution()

    def reducedCosts(self):
        return self.o.reduced_costs()

    def statusArray(self):
        return self.o.status()

    def pivotVariable(self):
        return self.o.pivot_variable()

    def pivotLog(self):
        return self.o.pivot_log()

    def rowWeights(self):
        return self.o.row_weights()

    def stats(self):
        d = collections.defaultdict(int)
        d.update(refactorizations=self.o.refactorizations, lu_factorizations=1, lu_active=1, lu_front=1, lu_tail=1, refreshes=100)
        return d


def main():
    torch.cuda.is_available = lambda: True
    E.ClpGpuSimplex = StandIn
    os.makedirs(OUT, exist_ok=True)
    keep = O._source_hash()
    before = set(os.listdir(OUT))
    sel = sys.argv[1:] or [os.path.join(ROOT, "tests")]
    pytest.main(["-m", "gpu", "-q", "-x" if False else "--tb=no", "-p", "no:cacheprovider", *sel])
    after = set(os.listdir(OUT))
    print(f"oracle source {keep[:12]}: {len(after - before)} new record(s), {len(after)} in {OUT}")
    if os.environ.get("CLP_ORACLE_REKEY_FROM") and len(after - before) == len(before):
        for name in before:  # every old record has its copy under the new name: the stale ones go
            os.remove(os.path.join(OUT, name))
        print(f"removed the {len(before)} records of source {os.environ['CLP_ORACLE_REKEY_FROM'][:12]}")


if __name__ == "__main__":
    main()
