"""ClpSimplexProgress::looping (src/ClpSolve.cpp:4438-4611, dual branch) three ways on the CPU: the oracle's restatement
(orc_test_looping), the engine's host restatement (clpgpu_test_looping -- host code of libclpgpu.so, no device involved) and a
Python restatement written from the reference for this test.  Crafted histories walk every branch: nothing before ten checks,
a repeat at a different iteration, the "only the last one matched" and "progress flag" exemptions, "really stuck", the first
action (tolerance x 1.05, dual bound x 1.1), the flagging of the newest incoming variable, all flagged (4), and the tenth bad
time (victory 0 / give up 3)."""
import struct

import numpy as np
import pytest

PROGRESS = 5


def same_bits(a, b):
    return struct.pack("d", a) == struct.pack("d", b)


def reference_looping(objective, infeasibility, count, iteration, flag_bits, newest):
    """ClpSolve.cpp:4438-4611 for algorithm_ < 0, state as in ClpSimplexProgress::reset :4613"""
    obj = [-1.7976931348623157e308 * 1.0e-50] * PROGRESS
    inf = [-1.0] * PROGRESS
    num = [-1] * PROGRESS
    its = [-1] * PROGRESS
    times = bad = 0
    tolerance, bound, force = 1.0e-7, 1.0e10, -1
    out = []
    for k in range(len(objective)):
        o, f, c, it = float(objective[k]), float(infeasibility[k]), int(count[k]), int(iteration[k])
        matched_bits = number_matched = nsame = 0
        for i in range(PROGRESS):
            if same_bits(o, obj[i]) and same_bits(f, inf[i]) and c == num[i]:
                matched_bits |= 1 << i
                if it != its[i]:
                    number_matched += 1
                else:
                    nsame += 1
            if i:
                obj[i - 1], inf[i - 1], num[i - 1], its[i - 1] = obj[i], inf[i], num[i], its[i]
        obj[-1], inf[-1], num[-1], its[-1] = o, f, c, it
        if nsame == PROGRESS:
            number_matched = PROGRESS
        if flag_bits[k] & 3:
            number_matched = 0
        times += 1
        if times < 10:
            number_matched = 0
        if matched_bits == 1 << (PROGRESS - 1):
            number_matched = 0
        code, flagged = -1, -1
        if number_matched:
            bad += 1
            if bad < 10:
                force = 1
                code = -2
                if bad < 2:
                    tolerance *= 1.05
                    if bound < 1.0e17:
                        bound *= 1.1
                else:
                    if bound > 1.0e14:
                        bound = 1.0e14
                    if newest[k] >= 0:
                        flagged = int(newest[k])
                        bad = 2
                    else:
                        code = 4
            else:
                code = 0 if f < 1.0e-4 else 3
        out.append((code, tolerance, bound, force, flagged))
    return out


def history(kind, rng):
    n = 40
    objective = np.cumsum(rng.uniform(0.5, 2.0, n))
    infeasibility = rng.uniform(1.0, 5.0, n)
    count = rng.integers(1, 50, n).astype(np.int32)
    iteration = (np.arange(n) * 40 + 40).astype(np.int32)
    flags = np.zeros(n, np.int32)
    newest = rng.integers(0, 64, n).astype(np.int32)
    if kind == "healthy":
        pass
    elif kind == "stall":  # the same triple from check 12 on: repeats at different iterations
        objective[12:] = objective[12]
        infeasibility[12:] = infeasibility[12]
        count[12:] = count[12]
    elif kind == "stall_small_infeasibility":  # nothing to flag and a tiny infeasibility: ends in "declare victory"
        objective[12:] = objective[12]
        infeasibility[12:] = 1.0e-6
        count[12:] = 1
        newest[:] = -1
    elif kind == "stall_with_progress_flags":  # every other check had a fixed variable leave
        objective[12:] = objective[12]
        infeasibility[12:] = infeasibility[12]
        count[12:] = count[12]
        flags[12::2] = 1
    elif kind == "early_stall":  # repeats before ten checks were made are ignored
        objective[2:9] = objective[2]
        infeasibility[2:9] = infeasibility[2]
        count[2:9] = count[2]
    elif kind == "really_stuck":  # no pivots at all between checks
        objective[12:] = objective[12]
        infeasibility[12:] = infeasibility[12]
        count[12:] = count[12]
        iteration[12:] = iteration[12]
    elif kind == "all_flagged":  # nothing left to flag
        objective[12:] = objective[12]
        infeasibility[12:] = infeasibility[12]
        count[12:] = count[12]
        newest[:] = -1
    elif kind == "alternating":  # period two: the check before last matches
        objective[12::2] = objective[12]
        objective[13::2] = objective[13]
        infeasibility[12::2] = infeasibility[12]
        infeasibility[13::2] = infeasibility[13]
        count[12::2] = count[12]
        count[13::2] = count[13]
    elif kind == "huge_bound":  # dual bound already beyond 1e17: not enlarged, later capped at 1e14
        objective[12:] = objective[12]
        infeasibility[12:] = infeasibility[12]
        count[12:] = count[12]
    return objective, infeasibility, count, iteration, flags, newest


KINDS = ["healthy", "stall", "stall_small_infeasibility", "stall_with_progress_flags", "early_stall", "really_stuck", "all_flagged", "alternating"]


@pytest.mark.parametrize("kind", KINDS)
def test_looping_three_ways(kind):
    from clp_amd import engine
    from oracle import oracle as orc

    engine.build()
    args = history(kind, np.random.default_rng(len(kind)))
    expected = reference_looping(*args)
    for name, got in (("oracle", orc.test_looping(*args)), ("engine", engine.test_looping(*args))):
        code, tol, bound, force, flagged = got
        for i, (c, t, b, f, fl) in enumerate(expected):
            assert (int(code[i]), float(tol[i]), float(bound[i]), int(force[i]), int(flagged[i])) == (c, t, b, f, fl), (name, kind, i)
    codes = [e[0] for e in expected]
    if kind in ("healthy", "early_stall"):
        assert set(codes) == {-1}
    if kind == "stall":
        # first the tolerances move, then one variable is flagged per check (which resets the bad-times count to 2, :4590)
        assert codes[:14] == [-1] * 14 and set(codes[14:]) == {-2}
        assert expected[14][4] == -1 and all(e[4] >= 0 for e in expected[15:])
        assert expected[14][1] == 1.0e-7 * 1.05 and expected[14][2] == 1.0e10 * 1.1 and expected[14][3] == 1
    if kind == "stall_small_infeasibility":
        assert 4 in codes and codes[-1] == 0
    if kind == "all_flagged":
        assert 4 in codes and codes[-1] == 3
    if kind == "stall_with_progress_flags":
        assert -2 in codes and codes[12] == -1
