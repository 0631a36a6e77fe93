"""Minimal MPS reader (fixed or free format) -> column-ordered LP arrays.

Plays the role CoinMpsIO plays for ``ClpModel::readMps`` (reference: src/ClpModel.cpp readMps ->
CoinMpsIO, out of tree).  Only what the dual-simplex path needs: ROWS / COLUMNS / RHS / RANGES /
BOUNDS, one objective row, minimisation.  Returns plain numpy arrays in the layout
``ClpPackedMatrix`` uses (CSC: columnStart int32, row int32, element float64).
"""
from __future__ import annotations

import numpy as np

INF = 1.0e30  # Clp treats |bound| >= 1e30 as infinite (COIN_DBL_MAX in the model arrays)


class LpData(dict):
    """dict with attribute access: m, n, col_start, row, elem, col_lower, col_upper, obj,
    row_lower, row_upper, (row_names, col_names, name, obj_offset)."""

    __getattr__ = dict.__getitem__

    def __setattr__(self, k, v):
        self[k] = v


def read_mps(path: str) -> LpData:
    rows: dict[str, int] = {}
    row_type: list[str] = []
    row_names: list[str] = []
    obj_name = None
    cols: dict[str, int] = {}
    col_names: list[str] = []
    entries: list[list[tuple[int, float]]] = []
    obj: list[float] = []
    rhs: dict[int, float] = {}
    ranges: dict[int, float] = {}
    bounds: list[tuple[str, int, float]] = []
    obj_offset = 0.0
    name = ""
    section = None
    with open(path) as fh:
        for raw in fh:
            if not raw.strip() or raw[0] == "*":
                continue
            if raw[0] not in " \t":
                tok = raw.split()
                section = tok[0].upper()
                if section == "NAME" and len(tok) > 1:
                    name = tok[1]
                if section == "ENDATA":
                    break
                continue
            tok = raw.split()
            if section == "ROWS":
                t, r = tok[0].upper(), tok[1]
                if t == "N":
                    if obj_name is None:
                        obj_name = r
                    continue
                rows[r] = len(row_names)
                row_names.append(r)
                row_type.append(t)
            elif section == "COLUMNS":
                if len(tok) >= 3 and tok[1] == "'MARKER'":
                    continue
                c = tok[0]
                if c not in cols:
                    cols[c] = len(col_names)
                    col_names.append(c)
                    entries.append([])
                    obj.append(0.0)
                j = cols[c]
                for k in range(1, len(tok) - 1, 2):
                    r, v = tok[k], float(tok[k + 1])
                    if r == obj_name:
                        obj[j] = v
                    elif r in rows:
                        entries[j].append((rows[r], v))
            elif section == "RHS":
                start = 1 if len(tok) % 2 == 1 else 0
                for k in range(start, len(tok) - 1, 2):
                    r, v = tok[k], float(tok[k + 1])
                    if r == obj_name:
                        obj_offset = -v
                    elif r in rows:
                        rhs[rows[r]] = v
            elif section == "RANGES":
                start = 1 if len(tok) % 2 == 1 else 0
                for k in range(start, len(tok) - 1, 2):
                    r, v = tok[k], float(tok[k + 1])
                    if r in rows:
                        ranges[rows[r]] = v
            elif section == "BOUNDS":
                t = tok[0].upper()
                if t in ("FR", "MI", "PL", "BV"):
                    c = tok[2] if len(tok) >= 3 else tok[1]
                    bounds.append((t, cols[c], 0.0))
                else:
                    if len(tok) == 4:
                        c, v = tok[2], float(tok[3])
                    else:
                        c, v = tok[1], float(tok[2])
                    bounds.append((t, cols[c], v))
    m, n = len(row_names), len(col_names)
    row_lower = np.full(m, -INF)
    row_upper = np.full(m, INF)
    for i, t in enumerate(row_type):
        b = rhs.get(i, 0.0)
        if t == "E":
            row_lower[i] = row_upper[i] = b
        elif t == "L":
            row_upper[i] = b
        elif t == "G":
            row_lower[i] = b
        if i in ranges:
            r = ranges[i]
            if t == "E":
                if r >= 0:
                    row_upper[i] = b + r
                else:
                    row_lower[i] = b + r
            elif t == "L":
                row_lower[i] = b - abs(r)
            elif t == "G":
                row_upper[i] = b + abs(r)
    col_lower = np.zeros(n)
    col_upper = np.full(n, INF)
    for t, j, v in bounds:
        if t == "UP":
            col_upper[j] = v
            if v < 0 and col_lower[j] == 0.0:
                col_lower[j] = -INF
        elif t == "LO":
            col_lower[j] = v
        elif t == "FX":
            col_lower[j] = col_upper[j] = v
        elif t == "FR":
            col_lower[j], col_upper[j] = -INF, INF
        elif t == "MI":
            col_lower[j] = -INF
        elif t == "PL":
            col_upper[j] = INF
        elif t == "BV":
            col_lower[j], col_upper[j] = 0.0, 1.0
    col_start = np.zeros(n + 1, dtype=np.int32)
    for j in range(n):
        entries[j].sort()
        col_start[j + 1] = col_start[j] + len(entries[j])
    row = np.fromiter((r for e in entries for r, _ in e), dtype=np.int32, count=int(col_start[n]))
    elem = np.fromiter((v for e in entries for _, v in e), dtype=np.float64, count=int(col_start[n]))
    return LpData(name=name, m=m, n=n, col_start=col_start, row=row, elem=elem, col_lower=col_lower,
                  col_upper=col_upper, obj=np.asarray(obj, dtype=np.float64), row_lower=row_lower,
                  row_upper=row_upper, row_names=row_names, col_names=col_names, obj_offset=obj_offset)
