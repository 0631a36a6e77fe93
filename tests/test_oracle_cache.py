"""The committed records of the oracle's long solves (oracle/oracle.py, tests/golden/oracle_cache/): a recorded solve must answer every
getter with exactly what the live solve answers, anything a record cannot answer must replay the real solve, and a record is only ever
found under the hash of the oracle's source, the LP's arrays, the options and the starting statuses."""
import os

import numpy as np

from clp_amd import problems as P
from oracle import oracle as O


def test_recorded_solve_equals_live_solve(tmp_path, monkeypatch):
    monkeypatch.setattr(O, "_CACHE_DIR", str(tmp_path))
    monkeypatch.setattr(O, "_CACHE_WRITE", str(tmp_path))
    monkeypatch.setattr(O, "_CACHE_MIN_S", 0.0)
    lp = P.sparse_lp(200, 800, 6, 3)

    def solve(**opts):
        o = O.OracleSimplex(lp)
        o.set_option("pivot_rule", 1)
        for k, v in opts.items():
            o.set_option(k, v)
        return o, o.dual()

    live, code = solve()
    assert live._rec is None and code == 0 and len(os.listdir(tmp_path)) == 1
    again, code2 = solve()
    assert again._rec is not None and code2 == code
    for name in ("solution", "reduced_costs", "status", "pivot_variable", "row_duals", "pivot_log"):
        assert np.array_equal(getattr(live, name)(), getattr(again, name)()), name
    assert np.array_equal(live.row_weights()[0], again.row_weights()[0]) and np.array_equal(live.row_weights()[1], again.row_weights()[1])
    assert (live.iterations, live.refactorizations, live.objective) == (again.iterations, again.refactorizations, again.objective)
    # a recorded solve has no clock of this machine: `seconds` refuses, `recorded_seconds` names the authoring box's; live=True solves here
    import pytest

    with pytest.raises(RuntimeError):
        again.seconds
    assert again.recorded_seconds is not None and live.recorded_seconds is None and live.seconds >= 0.0
    timed = O.OracleSimplex(lp)
    timed.set_option("pivot_rule", 1)
    assert timed.has_record() and timed.dual(live=True) == code and timed._rec is None and timed.seconds >= 0.0
    # another option set is another solve: no record, and it is written
    other, _ = solve(max_pivots=7)
    assert other._rec is None and len(os.listdir(tmp_path)) == 2
    # what a record cannot answer replays the real solve first: a unit-level call, and a second dual() (a warm re-solve)
    e = np.zeros(lp.m)
    e[5] = 1.0
    assert np.array_equal(again.ftran(e), live.ftran(e)) and again._rec is None
    third, _ = solve()
    assert third._rec is not None
    assert third.dual() == live.dual() and third.iterations == live.iterations


def test_committed_records_belong_to_the_current_oracle_source():
    """a record written for another oracle source can never be read (its key cannot be formed any more); the generator reports
    how many it wrote -- this only says the directory is not full of dead files"""
    d = O._CACHE_DIR
    if not os.path.isdir(d) or not os.listdir(d):
        return
    with np.load(os.path.join(d, sorted(os.listdir(d))[0])) as z:
        assert {"counters", "scalars", "solution", "pivot_log"} <= set(z.files)
