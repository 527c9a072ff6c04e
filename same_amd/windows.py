"""Sliding-window tiler: the grid, merge-right / merge-down and central-trim rules of
sliding_window_matching (src/same.py:481-488, :509-582) as a plan of independent windows.

Windows are the shard unit for very large sections (BASELINE config 5).  Cell counts of every
box the merge logic can ask about (base, right-merged, down-merged, both) come from one
batched device pass (csrc/sweep.hip window_count_kernel) instead of one pandas boolean mask
per window; the sequential walk that decides merges is then pure host lookups."""
import numpy as np

from . import ops


def window_grid(ref_xy, mov_xy, window_size, overlap):
    """src/same.py:481-488."""
    x_min = min(ref_xy[:, 0].min(), mov_xy[:, 0].min())
    x_max = max(ref_xy[:, 0].max(), mov_xy[:, 0].max())
    y_min = min(ref_xy[:, 1].min(), mov_xy[:, 1].min())
    y_max = max(ref_xy[:, 1].max(), mov_xy[:, 1].max())
    step = window_size - overlap
    xs = list(range(int(x_min), int(x_max), step))
    ys = list(range(int(y_min), int(y_max), step))
    return xs, ys, (x_min, x_max, y_min, y_max)


def _candidate_boxes(xs, ys, ws):
    """All boxes the walk may query: variant 0 base, 1 right-merged, 2 down-merged, 3 both."""
    boxes = np.empty((len(xs), len(ys), 4, 4), np.float64)
    for i, x in enumerate(xs):
        xr = xs[i + 1] + ws if i + 1 < len(xs) else x + ws
        for j, y in enumerate(ys):
            yd = ys[j + 1] + ws if j + 1 < len(ys) else y + ws
            boxes[i, j, 0] = (x, x + ws, y, y + ws)
            boxes[i, j, 1] = (x, xr, y, y + ws)
            boxes[i, j, 2] = (x, x + ws, y, yd)
            boxes[i, j, 3] = (x, xr, y, yd)
    return boxes


def window_plan(ref_xy, mov_xy, window_size, overlap, min_cells, skip_ids=(), ctx=None):
    """The windows the reference's double while-loop would run, in order.

    Each entry: i0/j0 (grid cell where the window starts), i/j (after merges), grid_id
    (= len(xs)*j0+i0, what the resume check compares), window_id (= len(xs)*j+i, what is stored),
    box (x0,x1,y0,y1 half-open), trim (central region kept), n_ref, n_mov."""
    ref_xy = np.ascontiguousarray(ref_xy, dtype=np.float64).reshape(-1, 2)
    mov_xy = np.ascontiguousarray(mov_xy, dtype=np.float64).reshape(-1, 2)
    xs, ys, (x_min, x_max, y_min, y_max) = window_grid(ref_xy, mov_xy, window_size, overlap)
    if not xs or not ys:
        return []
    boxes = _candidate_boxes(xs, ys, window_size)
    flat = boxes.reshape(-1, 4)
    n_ref = ops.window_count(ref_xy, flat, ctx=ctx).reshape(boxes.shape[:3])
    n_mov = ops.window_count(mov_xy, flat, ctx=ctx).reshape(boxes.shape[:3])
    skip_ids = set(skip_ids)

    plan = []
    i = 0
    while i < len(xs):
        j = 0
        while j < len(ys):
            if len(xs) * j + i in skip_ids:
                j += 1
                continue
            i0, j0 = i, j
            x, y = xs[i], ys[j]
            variant = 0
            nr, nm = n_ref[i0, j0, 0], n_mov[i0, j0, 0]
            if nr < min_cells or nm < min_cells:
                if i + 1 < len(xs):
                    variant = 1
                    nr, nm = n_ref[i0, j0, 1], n_mov[i0, j0, 1]
                    if nr >= min_cells and nm >= min_cells:
                        i += 1
                if (nr < min_cells or nm < min_cells) and j + 1 < len(ys):
                    variant |= 2
                    nr, nm = n_ref[i0, j0, variant], n_mov[i0, j0, variant]
                    if nr >= min_cells and nm >= min_cells:
                        j += 1
            if nr >= min_cells and nm >= min_cells:
                x0, x1, y0, y1 = (float(v) for v in boxes[i0, j0, variant])
                left, right = x == int(x_min), x1 >= int(x_max)
                top, bottom = y == int(y_min), y1 >= int(y_max)
                trim = (x0 if left else x0 + overlap / 2, x1 if right else x1 - overlap / 2,
                        y0 if top else y0 + overlap / 2, y1 if bottom else y1 - overlap / 2)
                plan.append({"i0": i0, "j0": j0, "i": i, "j": j, "grid_id": len(xs) * j0 + i0,
                             "window_id": len(xs) * j + i, "box": (x0, x1, y0, y1), "trim": trim,
                             "n_ref": int(nr), "n_mov": int(nm)})
            j += 1
        i += 1
    return plan


def assign_windows(plan, n_ranks):
    """Round-robin windows over ranks, heaviest first (windows are independent: no collective)."""
    order = sorted(range(len(plan)), key=lambda w: -(plan[w]["n_ref"] * plan[w]["n_mov"]))
    shards = [[] for _ in range(n_ranks)]
    for pos, w in enumerate(order):
        shards[pos % n_ranks].append(w)
    return [sorted(s) for s in shards]
