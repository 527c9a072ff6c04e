"""Where does Qhull itself stop being exact?  same_delaunay2d (csrc/delaunay.cpp) reports, for a finished triangulation, the
smallest ratio between (the distance of an edge's far corner from the lifted triangle's plane, as Qhull's 'Qbb'-scaled paraboloid
coordinates have it) and (Qhull's round-off allowance for these coordinates): its `margin`.  This tool builds point sets whose
margins span many decades -- by moving generic sets away from the origin, which costs Qhull's uncentred lifted coordinate its
digits -- and compares scipy.spatial.Delaunay's simplices with the exact triangulation (same_delaunay2d with guard = 0: its signs
are computed on centred differences and do not care about the offset).  Printed per decade of the margin: sets, sets whose scipy
triangles differ.  The largest margin at which a difference is seen calibrates delaunay.GUARD (16: 60 x it).
CPU only.  Usage: python3 tools/delaunay_margin.py [seed=11] [sets=500]"""
import math
import os
import sys

import numpy as np
from scipy.spatial import Delaunay

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from same_amd import delaunay                      # noqa: E402


def canonical(t):
    t = np.sort(np.asarray(t, np.int64), axis=1)
    return t[np.lexsort((t[:, 2], t[:, 1], t[:, 0]))]


seed, n_sets = (int(sys.argv[1]) if len(sys.argv) > 1 else 11), (int(sys.argv[2]) if len(sys.argv) > 2 else 500)
rng = np.random.default_rng(seed)
kinds = ("uniform square", "anisotropic blob", "clusters", "strip")
rows = []
for _ in range(n_sets):
    n = int(rng.integers(50, 5000))
    off = 10.0 ** rng.uniform(2, 8.5) * rng.choice([-1, 1], 2) * rng.choice([0, 1, 1, 1], 2)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        pts = rng.uniform(0, 1200, (n, 2))
    elif kind == 1:
        pts = rng.normal(0, 1, (n, 2)) * rng.uniform(1, 300, 2)
    elif kind == 2:
        c = rng.uniform(0, 1200, (max(n // 40, 1), 2))
        pts = c[rng.integers(0, len(c), n)] + rng.normal(0, rng.uniform(0.5, 20), (n, 2))
    else:
        pts = np.c_[rng.uniform(0, 3000, n), rng.uniform(0, 30, n)]
    pts = pts + off
    exact, margin = delaunay.native_simplices(pts, guard=0.0, with_margin=True)
    if exact is None:                              # a sign in doubt on centred differences: a truly degenerate set
        rows.append((margin, None, kind))
        continue
    qhull = Delaunay(pts).simplices
    rows.append((margin, bool(len(exact) == len(qhull) and np.array_equal(canonical(exact), canonical(qhull))), kind))
bins = {}
for margin, same, kind in rows:
    d = bins.setdefault(math.floor(math.log10(max(margin, 1e-30))), [0, 0, 0])
    d[0] += 1
    d[1] += same is False
    d[2] += same is None
print(f"seed {seed}, {n_sets} sets (uniform squares, anisotropic blobs, clusters, strips; 50-5000 points; offsets 1e2..3e8, either sign)")
for b in sorted(bins):
    print(f"margin in [1e{b}, 1e{b + 1}): sets {bins[b][0]:4d}   scipy's triangles differ from the exact ones in {bins[b][1]:4d}   no "
          f"exact answer {bins[b][2]}")
worst = max(((m, kinds[k]) for m, same, k in rows if same is False), default=None)
print("largest margin at which scipy's triangles differ:", worst, "-- delaunay.GUARD =", delaunay.GUARD)
