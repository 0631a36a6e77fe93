"""Column-range sharding of row pricing across the GPUs of one node.

The reference's shared-memory precedent is ABOCA_LITE: ``chunk = (n + T - 1) / T`` contiguous column
ranges, per-chunk (numberNonZero, numberRemaining, upperTheta), then ``min(upperTheta)``, summed
counts and lists concatenated in chunk order (src/ClpPackedMatrix.cpp:1823-1854).  Rank-major
concatenation preserves the by-column candidate order, so ties in the ratio test resolve exactly as
on one GPU.
"""
from __future__ import annotations

import numpy as np

PRICE_BLOCK = 256  # the pricing kernel's workgroup covers 256 columns; keep shard edges aligned


def column_ranges(n: int, nranks: int):
    """The engine's shard of every rank (clpgpu_context::applyShard): equal chunks rounded up to the
    pricing workgroup's 256 columns, clipped at n -- trailing ranks of a small LP may be empty."""
    chunk = (n + nranks - 1) // nranks
    chunk = ((chunk + PRICE_BLOCK - 1) // PRICE_BLOCK) * PRICE_BLOCK
    return [(min(r * chunk, n), min((r + 1) * chunk, n)) for r in range(nranks)]


def merge_candidates(parts):
    """parts[r] = (out_index, out_value, cand_index, cand_value, upper_theta) of rank r."""
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), np.concatenate([p[3] for p in parts]),
            min(p[4] for p in parts))
