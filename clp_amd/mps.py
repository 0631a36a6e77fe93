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


def write_mps(lp, path: str) -> None:
    """Free-format MPS of `lp` (the counterpart of ClpModel::writeMps -> CoinMpsIO::writeMps, out of
    tree), so that a real `clp` binary can be timed on exactly the LP the engine solves
    (`clp file.mps -presolve off -dualpivot steepest -dualsimplex`).  Rows are
    written as E (equal bounds), L / G (one finite bound), G + RANGES (two finite bounds) or N-free
    rows turned into `G -1e30` never occur in the generators; values use repr-exact 17 digits so the
    LP read back is bit-identical.  Vectorised: the 10 M-nonzero bench LP takes a few tens of seconds."""
    m, n = int(lp.m), int(lp.n)
    rl, ru = np.asarray(lp.row_lower, float), np.asarray(lp.row_upper, float)
    cl, cu = np.asarray(lp.col_lower, float), np.asarray(lp.col_upper, float)
    obj = np.asarray(lp.obj, float)
    lo_inf, up_inf = rl <= -1e20, ru >= 1e20
    rtype = np.where(rl == ru, "E", np.where(lo_inf & ~up_inf, "L", np.where(~lo_inf & up_inf, "G", np.where(lo_inf & up_inf, "N", "G"))))
    rname = np.char.add("R", np.arange(m).astype(str))
    cname = np.char.add("C", np.arange(n).astype(str))
    with open(path, "w") as fh:
        fh.write(f"NAME {getattr(lp, 'name', '') or 'LP'}\nROWS\n N OBJ\n")
        fh.write("\n".join(np.char.add(np.char.add(np.char.add(" ", rtype), " "), rname)))
        fh.write("\nCOLUMNS\n")
        counts = np.diff(np.asarray(lp.col_start, np.int64))
        col_of = np.repeat(np.arange(n), counts)
        vals = np.char.mod("%.17g", np.asarray(lp.elem, float))
        body = np.char.add(np.char.add(np.char.add(np.char.add(" ", cname[col_of]), " "), rname[np.asarray(lp.row, np.int64)]), " ")
        body = np.char.add(body, vals)
        # objective entries first for each column that has one (order inside COLUMNS is free per column)
        has_obj = np.nonzero(obj != 0.0)[0]
        obj_lines = np.char.add(np.char.add(np.char.add(" ", cname[has_obj]), " OBJ "), np.char.mod("%.17g", obj[has_obj]))
        # interleave so that all lines of one column are contiguous (CoinMpsIO requires it)
        key = np.concatenate([col_of * 2 + 1, has_obj * 2])
        lines = np.concatenate([body, obj_lines])[np.argsort(key, kind="stable")]
        fh.write("\n".join(lines))
        fh.write("\nRHS\n")
        rhs = np.where(rtype == "L", ru, rl)
        sel = np.nonzero((rtype != "N") & (rhs != 0.0))[0]
        if len(sel):
            fh.write("\n".join(np.char.add(np.char.add(np.char.add(" RHS ", rname[sel]), " "), np.char.mod("%.17g", rhs[sel]))))
            fh.write("\n")
        rng = np.nonzero((rtype == "G") & ~up_inf & ~lo_inf)[0]
        if len(rng):
            fh.write("RANGES\n")
            fh.write("\n".join(np.char.add(np.char.add(np.char.add(" RNG ", rname[rng]), " "), np.char.mod("%.17g", ru[rng] - rl[rng]))))
            fh.write("\n")
        fh.write("BOUNDS\n")
        out = []
        free = (cl <= -1e20) & (cu >= 1e20)
        fixed = (cl == cu) & ~free
        for tag, sel, val in (("FR", free, None), ("FX", fixed, cl), ("MI", (cl <= -1e20) & ~free, None),
                              ("LO", (cl > -1e20) & (cl != 0.0) & ~fixed, cl), ("UP", (cu < 1e20) & ~fixed, cu)):
            idx = np.nonzero(sel)[0]
            if not len(idx):
                continue
            line = np.char.add(f" {tag} BND ", cname[idx])
            if val is not None:
                line = np.char.add(np.char.add(line, " "), np.char.mod("%.17g", val[idx]))
            out.append(line)
        if out:
            # MI before UP for the same column keeps "UP with negative value" readers happy
            fh.write("\n".join(np.concatenate(out)))
            fh.write("\n")
        fh.write("ENDATA\n")
