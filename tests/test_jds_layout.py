"""The jagged row-tiled layout k_price_lds prices from (clp_amd/csrc/engine.hip jdsLayout; device_state.h; the reference prices a
dense pi by column straight from its CSC copy, ClpPackedMatrix::transposeTimesByColumn, src/ClpPackedMatrix.cpp:961), walked
on the CPU exactly the way the kernel walks it -- slice by slice, tile after tile, a pair of entries per step, the active lanes a
prefix of the wave, the accumulators permuted between tiles -- with no device involved (clpgpu_test_jds_layout is host code):

  * every matrix entry of every placed column appears exactly once, in the column's own entry order;
  * each lane's running sum, taken with numpy's separately rounded multiply and add, is BIT-identical to the sequential dot
    product of that column with pi (the kernel's fused multiply-adds are held to the oracle on the GPU, tests/test_gpu_lu.py);
  * the stream is contiguous: each slice starts where the previous one ended, inactive halves of a pair are (row 0, 0.0).
"""
import numpy as np
import pytest

from clp_amd import problems as P

SELL_LONG = 128  # clp_amd/csrc/kernels.hip


def window_order(lp, holes=None):
    """buildSell's placement: inside each window of 256 columns, by decreasing length (stable); long columns leave"""
    cs = np.asarray(lp.col_start)
    lens = np.diff(cs)
    n = lp.n
    nwin = (n + 255) // 256
    order = np.full(nwin * 256, -1, np.int32)
    for w in range(nwin):
        cols = np.arange(w * 256, min(n, (w + 1) * 256))
        cols = cols[lens[cols] <= SELL_LONG]
        if holes is not None:
            cols = cols[~holes[cols]]
        cols = cols[np.argsort(-lens[cols], kind="stable")]
        order[w * 256:w * 256 + len(cols)] = cols
    return order


def walk(lp, order, lay, pi):
    cs, row, elem = np.asarray(lp.col_start), np.asarray(lp.row), np.asarray(lp.elem)
    T, R = lay["tiles"], lay["tile_rows"]
    assert T * R >= lp.m and R % 128 == 0 and R <= 16768
    pad = np.zeros(T * R)
    pad[:lp.m] = pi
    slices = order.size // 64
    lo16, hi16 = lay["row_pair"] & 0xFFFF, lay["row_pair"] >> 16
    e0, e1 = lay["elem_pair"][:, 0], lay["elem_pair"][:, 1]
    values = np.zeros(order.size)
    seen_rows = [[] for _ in range(order.size)]
    seen_elem = [[] for _ in range(order.size)]
    off = 0
    for s in range(slices):
        assert lay["seg_start"][s] == off, "slices follow each other in the record stream"
        acc = np.zeros(64)
        lane_home = np.arange(64)  # which home lane's sum sits at position p
        for tau in range(T):
            cnt = lay["cnt"][s, tau].astype(np.int64)
            if tau:
                src = lay["src"][s, tau]
                acc, lane_home = acc[src], lane_home[src]
            else:
                assert np.array_equal(lay["src"][s, 0], np.argsort(-cnt_home_first(lp, order, s, R), kind="stable"))
                acc, lane_home = acc[lay["src"][s, 0]], lane_home[lay["src"][s, 0]]
            assert np.all(np.diff(cnt) <= 0), "lanes are sorted by count: the active ones are a prefix"
            for q in range((int(cnt.max()) + 1) // 2):
                k = int(np.sum(cnt > 2 * q))
                rec = np.arange(off, off + k)
                full = cnt[:k] > 2 * q + 1
                assert np.all(lo16[rec] < R) and np.all(hi16[rec] < R)
                assert np.all(e1[rec][~full] == 0.0) and np.all(hi16[rec][~full] == 0)
                acc[:k] = acc[:k] + pad[tau * R + lo16[rec]] * e0[rec]
                # the second half of an odd column's last pair adds pi[tile start] * 0.0 in the kernel: exact, skipped here
                acc[:k][full] = acc[:k][full] + (pad[tau * R + hi16[rec]] * e1[rec])[full]
                for p in range(k):
                    h = s * 64 + lane_home[p]
                    seen_rows[h].append(tau * R + int(lo16[rec[p]]))
                    seen_elem[h].append(e0[rec[p]])
                    if full[p]:
                        seen_rows[h].append(tau * R + int(hi16[rec[p]]))
                        seen_elem[h].append(e1[rec[p]])
                off += k
        home = lay["home"][s]
        assert np.array_equal(np.sort(home), np.arange(64)), "the way home is a permutation"
        values[s * 64:(s + 1) * 64] = acc[home]
        assert np.array_equal(lane_home[home], np.arange(64))
    assert off == lay["records"]
    # the answer a sequential sweep gives (ClpPackedMatrix.cpp:961-1011: value += pi[row] * element, entry after entry)
    for i, j in enumerate(order):
        if j < 0:
            assert values[i] == 0.0 and not seen_rows[i]
            continue
        a, b = cs[j], cs[j + 1]
        assert seen_rows[i] == list(row[a:b]), f"column {j}: every entry once, in the column's order"
        assert seen_elem[i] == list(elem[a:b])
        v = 0.0
        for t in range(a, b):
            v = v + pi[row[t]] * elem[t]
        assert v == values[i], f"column {j}: bit-identical running sum"
    return values


def cnt_home_first(lp, order, s, R):
    cs, row = np.asarray(lp.col_start), np.asarray(lp.row)
    out = np.zeros(64, np.int64)
    for l in range(64):
        j = order[s * 64 + l]
        if j >= 0:
            out[l] = np.sum(row[cs[j]:cs[j + 1]] < R)
    return out


@pytest.mark.parametrize("shape", ["one_tile", "three_tiles", "long_columns_and_holes"])
def test_layout_walk_reproduces_every_column(built, shape):
    from clp_amd.engine import jds_layout

    rng = np.random.default_rng(5)
    holes = None
    if shape == "one_tile":
        lp = P.sparse_lp(4200, 1500, 7, seed=11)
    elif shape == "three_tiles":
        lp = P.sparse_lp(40000, 1100, 9, seed=12)  # 3 tiles of 13 440 rows
    else:
        lp = P.netlib_shaped_lp(20000, 1300, 30000, seed=13)  # power-law counts: some columns pass SELL_LONG and leave
        holes = rng.random(lp.n) < 0.1
        assert np.any(np.diff(lp.col_start) > SELL_LONG)
    order = window_order(lp, holes)
    lay = jds_layout(lp, order)
    assert lay is not None
    placed = order[order >= 0]
    lens = np.diff(lp.col_start)
    # a pair record per two entries of a (column, tile) run
    assert lay["records"] >= (int(lens[placed].sum()) + 1) // 2
    assert {"one_tile": 1, "three_tiles": 3}.get(shape, lay["tiles"]) == lay["tiles"]
    pi = rng.standard_normal(lp.m) * (rng.random(lp.m) < 0.7)
    walk(lp, order, lay, pi)


def test_layout_refuses_a_column_too_long_for_a_tile(built):
    """more than 255 entries of one column in one tile do not fit the byte counts: the layout says so (buildSell never offers
    such a column -- SELL_LONG = 128 -- and the engine then prices with the plain kernels)"""
    from clp_amd.engine import jds_layout

    lp = P.dense_lp(300, 64)
    order = np.arange(64, dtype=np.int32)
    assert jds_layout(lp, order) is None


@pytest.mark.parametrize("seed", range(4))
def test_layout_walk_on_ragged_shapes(built, seed):
    """column counts that are no multiple of the window or the slice, empty columns, a last window with a handful of columns, rows that
    leave the last tile nearly empty: the walk must still reproduce every column bit for bit"""
    from clp_amd.engine import jds_layout

    rng = np.random.default_rng(100 + seed)
    m = int(rng.choice([4097, 16769, 17000, 33600]))
    n = int(rng.integers(300, 900))
    lp = P.sparse_lp(m, n, int(rng.integers(2, 12)), seed=200 + seed)
    lp = type(lp)(lp)
    # empty some columns (their entries go, the starts close up)
    lens = np.diff(lp.col_start)
    drop = rng.random(n) < 0.05
    keep = np.repeat(~drop, lens)
    lp.row, lp.elem = lp.row[keep], lp.elem[keep]
    lp.col_start = np.concatenate([[0], np.cumsum(np.where(drop, 0, lens))]).astype(lp.col_start.dtype)
    order = window_order(lp, rng.random(n) < 0.03)
    lay = jds_layout(lp, order)
    assert lay is not None and lay["tiles"] == -(-m // 16768)
    walk(lp, order, lay, rng.standard_normal(m))
