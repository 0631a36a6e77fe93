"""The cycle detector (ClpSimplexProgress::cycle, src/ClpSolve.cpp:4726-4825, called from ClpSimplex::housekeeping :2397-2431).
No LP of the suite cycles on its own, so the detector is driven directly: a Python restatement of the reference function
(the third implementation, written from the reference text) is compared with the oracle's shifting-array form on the CPU
and with the device's ring-buffer form on the GPU, over sequences with a clean cycle, irregular repeats and none."""
import numpy as np
import pytest

CLP_CYCLE = 12


def reference_cycle(seq_in, seq_out, way_in, way_out):
    in_, out_, way_ = [-1] * CLP_CYCLE, [-1] * CLP_CYCLE, [0] * CLP_CYCLE
    res = []
    for a, b, wi, wo in zip(seq_in, seq_out, way_in, way_out):
        matched = 0
        for i in range(1, CLP_CYCLE):
            if a == out_[i]:
                matched = -1
                break
        if matched and in_[0] >= 0:
            matched, n_matched = 0, 0
            for k in range(1, CLP_CYCLE - 4):
                if (in_[0], out_[0], way_[0]) == (in_[k], out_[k], way_[k]):
                    n_matched += 1
                    end = CLP_CYCLE - k
                    j = 1
                    while j < end and (in_[j + k], out_[j + k], way_[j + k]) == (in_[j], out_[j], way_[j]):
                        j += 1
                    if j == end:
                        matched = k
                        break
            if matched <= 0 and n_matched > 1:
                matched = 100
        in_, out_, way_ = in_[1:] + [a], out_[1:] + [b], way_[1:] + [1 - wi + 4 * (1 - wo)]
        res.append(matched)
    return np.array(res, np.int32)


def sequences():
    rng = np.random.default_rng(3)
    out = []
    # a clean cycle of length 3 and of length 4, entered after some unrelated pivots
    for length in (2, 3, 4, 5):
        cyc = [(10 + i, 10 + (i + 1) % length, 1, -1) for i in range(length)]
        seq = [(100 + i, 200 + i, 1, 1) for i in range(7)] + cyc * 12
        out.append(seq)
    # irregular repeats of one pivot among others
    seq = []
    for i in range(40):
        seq.append((7, 9, 1, -1) if i % 3 == 0 else (300 + i, 9 if i % 5 == 0 else 400 + i, -1, 1))
    out.append(seq)
    # random traffic over a small set of variables (many accidental matches of `in` with an earlier `out`)
    out.append([(int(rng.integers(0, 6)), int(rng.integers(0, 6)), int(rng.choice([-1, 1])), int(rng.choice([-1, 1]))) for _ in range(300)])
    # nothing repeats
    out.append([(i, 1000 + i, 1, 1) for i in range(50)])
    return out


@pytest.mark.parametrize("case", range(7))
def test_oracle_cycle_detector_follows_the_reference(built, case):
    from oracle import oracle as orc

    seq = np.array(sequences()[case], np.int32)
    ref = reference_cycle(*seq.T)
    got = orc.test_cycle(*seq.T)
    assert np.array_equal(got, ref)
    if case < 4:
        assert ref.max() == case + 2  # the clean cycle is found with its length
    if case == 6:
        assert ref.max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(7))
def test_device_cycle_detector_follows_the_reference(built, case):
    import torch

    assert torch.cuda.is_available()
    from clp_amd import problems as P
    from clp_amd.engine import ClpGpuSimplex

    g = ClpGpuSimplex(0).loadProblem(P.sparse_lp(300, 1200, 8, seed=11))
    seq = np.array(sequences()[case], np.int32)
    assert np.array_equal(g.testCycle(*seq.T), reference_cycle(*seq.T))
