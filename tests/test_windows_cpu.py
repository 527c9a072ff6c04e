"""Host pieces of the window pipelines that need no GPU: the grid-binned box query both pipelines subset with, and the
column container."""
import numpy as np
import pandas as pd
import pytest


def test_grid_rows_equal_the_four_comparisons():
    """windows.GridRows.rows == np.flatnonzero of subset_data's comparisons (src/same.py:293-295) for boxes inside, across and
    outside the point set, degenerate boxes, points on box edges, NaN / infinite coordinates and degenerate point sets."""
    from same_amd.windows import GridRows

    rng = np.random.default_rng(0)
    sets = [rng.uniform(-50, 1050, (20000, 2)), rng.integers(0, 40, (5000, 2)).astype(float) * 25.0,      # lattice: points ON box edges
            np.column_stack((rng.uniform(0, 100, 300), np.full(300, 7.0))),                             # collinear (zero extent in y)
            np.tile([[3.0, 4.0]], (50, 1)), np.zeros((0, 2))]                                             # one location; empty
    sets[0][::97] = np.nan
    sets[0][5::101, 1] = np.inf
    for pts in sets:
        g = GridRows(pts[:, 0], pts[:, 1])
        boxes = [(0, 1000, 0, 1000), (100, 125, 100, 125), (-1e9, 1e9, -1e9, 1e9), (500, 500, 0, 1000), (1040, 2000, -100, 3), (25, 50, 25, 50),
                 (3, 3.0000001, 4, 4.0000001), (7, 3, 0, 10)] + [tuple(np.sort(rng.uniform(-100, 1100, 2))) + tuple(np.sort(rng.uniform(-100, 1100, 2)))
                                                                 for _ in range(40)]
        for x0, x1, y0, y1 in boxes:
            with np.errstate(invalid="ignore"):
                want = np.flatnonzero((pts[:, 0] >= x0) & (pts[:, 0] < x1) & (pts[:, 1] >= y0) & (pts[:, 1] < y1))
            got = g.rows(x0, x1, y0, y1)
            assert np.array_equal(got, want), (len(pts), x0, x1, y0, y1)
            rows, pos = g.positions(x0, x1, y0, y1, pts[g.order, 0], pts[g.order, 1])     # the same query on grid-ordered storage
            assert np.array_equal(rows, want) and np.array_equal(g.order[pos], rows)


def test_section_from_frame_keeps_what_the_pipeline_reads():
    from same_amd.windows import Section

    df = pd.DataFrame({"X": [1.0, 2.0, 3.0], "Y": [4.0, 5.0, 6.0], "cell_type": ["b", "a", "b"], "c1": [0.1, 0.2, 0.3], "c2": [9, 8, 7],
                       "size": np.array([1, 2, 3], np.int32)}, index=[10, 11, 12])
    s = Section.from_frame(df, ["c2", "c1"])
    assert s.xy.flags.c_contiguous and s.types.flags.c_contiguous and s.types.tolist() == [[9.0, 0.1], [8.0, 0.2], [7.0, 0.3]]
    assert s.type_id[0] == s.type_id[2] != s.type_id[1] and s.size.dtype == np.int32
    assert s.grid.rows(0, 2.5, 0, 10).tolist() == [0, 1]
    rows, pos = s.window((0, 2.5, 0, 10))
    assert rows.tolist() == [0, 1] and s.g_types[pos].tolist() == s.types[rows].tolist() and s.g_size[pos].tolist() == [1, 2]
    bare = Section.from_frame(df.drop(columns=["cell_type", "size"]), ["c1"])
    assert bare.type_id is None and bare.size.tolist() == [1, 1, 1] and np.issubdtype(bare.size.dtype, np.integer)


def test_window_cell_grid_sizes():
    """The grid the device sections are binned on: gcd(step, window) cells when a window is at most 5 x 5 of them (every box of the plan a
    union of cells), quarter windows otherwise."""
    from same_amd.windows import window_cell_grid

    assert window_cell_grid(([7, 907], [3, 903]), 1200, 300) == (7.0, 3.0, 300.0)
    assert window_cell_grid(([0], [0]), 1000, 250) == (0.0, 0.0, 250.0)
    assert window_cell_grid(([0], [0]), 1001, 250) == (0.0, 0.0, 1001 / 4.0)       # gcd 1
    assert window_cell_grid(([0], [0]), 700, 200) == (0.0, 0.0, 700 / 4.0)        # gcd 100: 7 x 7 cells a window
