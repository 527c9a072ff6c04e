"""Generate tests/golden/*.npz by RUNNING THE REFERENCE (build container only).

Run:  python3 -B tools/gen_golden.py            (the pre-MIP fixtures: synthetic_example, cfg1_500, cfg2_small, simulated_*, adversarial)
      python3 -B tools/gen_golden.py <family>   (eval | metacell | unpack | eager | runsame | tiler | sweep | tongue | heart)
      python3 -B tools/gen_golden.py all        (everything; ~8 min, dominated by the reference's own Python loops)
Needs /root/reference; never runs on the GPU box.  Fixtures are data only: the inputs and
the outputs the reference's own functions produced for them.  Functions that exist in the
reference are called as-is (find_knn_within_radius, filter_triangles_by_radius,
precompute_triangle_info, verify_spatial_preservation, compute_mip_start_pairs,
calculate_signed_area, calc_ref_area, find_knn_with_cell_type_priority).  run_same's inline
pre-MIP / sweep loops cannot be imported (they need a live gurobipy), so they
are driven here on the reference's own frames with the expression order of the cited lines.
"""
import contextlib
import io
import json
import os
import pickle
import sys

import numpy as np
import pandas as pd
from scipy.spatial import Delaunay

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_loader import REF_ROOT, load_reference  # noqa: E402

import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location("same_synth", os.path.join(ROOT, "same_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

OUT = os.environ.get("SAME_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # the test that re-generates into a scratch directory sets it
if len(sys.argv) > 1 and sys.argv[1] in ("eager", "runsame", "tiler", "tongue", "heart", "sweep"):
    # solver-facing fixtures: the reference's model builders / run_same talk to the recording solver double of the tests
    # (gurobipy itself is proprietary and absent), installed as `gurobipy` BEFORE the reference modules bind its names
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import fake_gurobipy as _fg
    _fg.install()
ref = load_reference(with_run_same=len(sys.argv) > 1 and sys.argv[1] in ("runsame", "tiler", "tongue", "heart", "sweep"))


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


# ------------------------------------------------------------------ inline loops of run_same
def ref_pair_costs(aligned_df, ref_df, valid_pairs, commonCT, dist_ct_coeff):
    """src/same.py:1180-1189."""
    c = []
    dist_coeff = dist_ct_coeff * 0.001
    for idx, (i, j) in enumerate(valid_pairs):
        dist_ct = np.sum(np.abs(aligned_df.iloc[i][commonCT] - ref_df.iloc[j][commonCT]))
        dist_coords = np.abs(aligned_df.iloc[i]['X'] - ref_df.iloc[j]['X']) + \
            np.abs(aligned_df.iloc[i]['Y'] - ref_df.iloc[j]['Y'])
        c.append(dist_ct_coeff * dist_ct + dist_coeff * dist_coords)
    return np.array(c, dtype=np.float64)


def ref_weights_signs(aligned_df, tris):
    """src/same.py:1128-1146."""
    w, s = [], []
    for tri in tris:
        a, b, c = tri
        w.append(aligned_df.iloc[a]['size'] + aligned_df.iloc[b]['size'] + aligned_df.iloc[c]['size'])
        ax, ay = aligned_df.iloc[a]['X'], aligned_df.iloc[a]['Y']
        bx, by = aligned_df.iloc[b]['X'], aligned_df.iloc[b]['Y']
        cx, cy = aligned_df.iloc[c]['X'], aligned_df.iloc[c]['Y']
        s.append(np.sign((bx - ax) * (cy - ay) - (by - ay) * (cx - ax)))
    return np.array(w, dtype=np.float64), np.array(s, dtype=np.float64)


def ref_lazy_sweep(x_vals, valid_pairs, tris, source_signs, ref_df):
    """src/same.py:634-669 (the arithmetic body of _lazy_orientation_callback)."""
    ref_coords = {j: (ref_df.iloc[j]['X'], ref_df.iloc[j]['Y']) for j in range(len(ref_df))}  # same.py:1158
    matching = {}
    for idx, (ip, jp) in enumerate(valid_pairs):
        if x_vals[idx] > 0.5:
            matching[ip] = jp
    violating, checked = [], 0
    for tri_idx, tri in enumerate(tris):
        a, b, c = tri
        if a not in matching or b not in matching or c not in matching:
            continue
        ax, ay = ref_coords[matching[a]]
        bx, by = ref_coords[matching[b]]
        cx, cy = ref_coords[matching[c]]
        ref_sign = np.sign((bx - ax) * (cy - ay) - (by - ay) * (cx - ax))
        source_sign = source_signs[tri_idx]
        if source_sign == 0 or ref_sign == 0:
            continue
        checked += 1
        if source_sign != ref_sign:
            violating.append(tri_idx)
    return checked, np.array(violating, dtype=np.int64)


def ref_area_flips(aligned_df, ref_df, tris, aligned_to_ref):
    """src/same.py:1362-1402 with helpers.calculate_signed_area."""
    before, after, flipped, m3 = [], [], [], []
    for tri_idx, triangle in enumerate(tris):
        p1, p2, p3 = triangle
        coords = [(aligned_df.iloc[p]['X'], aligned_df.iloc[p]['Y']) for p in (p1, p2, p3)]
        before.append(ref.helpers.calculate_signed_area(*coords))
    for tri_idx, triangle in enumerate(tris):
        matched = [p in aligned_to_ref for p in triangle]
        m3.append(matched)
        if not all(matched):
            after.append(np.nan)
            continue
        rc = [(ref_df.iloc[aligned_to_ref[p]]['X'], ref_df.iloc[aligned_to_ref[p]]['Y']) for p in triangle]
        area = ref.helpers.calculate_signed_area(rc[0], rc[1], rc[2])
        after.append(area)
        if before[tri_idx] * area < 0:
            flipped.append(tri_idx)
    return np.array(before), np.array(after), np.array(flipped, dtype=np.int64), np.array(m3, dtype=np.uint8)


def simplex_map(n, tris):
    """src/same.py:1096-1099."""
    m = {i: set() for i in range(n)}
    for idx, simplex in enumerate(tris):
        for i in simplex:
            m[i].add(idx)
    return m


def violations_to_arrays(v):
    def rows(lst):
        return np.array([(d['triangle_idx'], d['point1']['aligned_idx'], d['point2']['aligned_idx'],
                          d['point1']['ref_idx'], d['point2']['ref_idx']) for d in lst], dtype=np.int64).reshape(-1, 5)
    s = v['violation_summary']
    return {
        'viol_x': rows(v['x_order_violations']), 'viol_y': rows(v['y_order_violations']),
        'viol_tris': np.array(sorted(int(t) for t in v['triangles_with_violations']), dtype=np.int64),
        'viol_points': np.array(sorted(int(p) for p in v['points_with_violations']), dtype=np.int64),
        'viol_summary': np.array([s['total_triangles'], s['violated_triangles'], s['total_comparisons'],
                                  s['total_violations']], dtype=np.int64),
        'viol_percent': np.array([s['percent_triangles_violated'], s['percent_violations']], dtype=np.float64),
    }


def frame_arrays(prefix, df, commonCT):
    return {f'{prefix}_xy': df[['X', 'Y']].to_numpy(dtype=np.float64),
            f'{prefix}_types': df[list(commonCT)].to_numpy(dtype=np.float64),
            f'{prefix}_cell_type': df['cell_type'].to_numpy().astype(str),
            f'{prefix}_size': df['size'].to_numpy(dtype=np.float64)}


def full_case(name, aligned_in, ref_in, commonCT, radius, knn, min_angle_deg, dist_ct_coeff, no_match_penalty,
              cost_sample=None, seed=0, do_hungarian=True):
    """One end-to-end pass of the pre-MIP path + sweeps through the reference, saved as a fixture."""
    print(f"[{name}] n_mov={len(aligned_in)} n_ref={len(ref_in)} T={len(commonCT)} r={radius} k={knn}")
    out = {'commonCT': np.array(commonCT), 'params': np.array([radius, knn, -1 if min_angle_deg is None else min_angle_deg,
                                                                 dist_ct_coeff, no_match_penalty], dtype=np.float64)}
    out.update(frame_arrays('in_aligned', aligned_in, commonCT))
    out.update(frame_arrays('in_ref', ref_in, commonCT))

    # a2 -------------------------------------------------------------------------------
    a_df, r_df, pairs = quiet(ref.utils.find_knn_within_radius, aligned_in, ref_in, radius, knn=knn)
    pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
    out['pairs'] = pairs
    out['kept_aligned'] = a_df['__row'].to_numpy(dtype=np.int64)
    out['kept_ref'] = r_df['__row'].to_numpy(dtype=np.int64)
    print(f"  pairs={len(pairs)} aligned_kept={len(a_df)} ref_kept={len(r_df)}")

    # a3 -------------------------------------------------------------------------------
    _, _, prio = quiet(ref.knn_utils.find_knn_with_cell_type_priority, aligned_in, ref_in, radius, knn=knn)
    out['pairs_priority'] = np.asarray(prio, dtype=np.int64).reshape(-1, 2)

    # a4 -------------------------------------------------------------------------------
    if cost_sample is not None and cost_sample < len(pairs):
        sel = np.sort(np.random.default_rng(seed).choice(len(pairs), cost_sample, replace=False))
    else:
        sel = np.arange(len(pairs))
    out['cost_sel'] = sel
    out['costs'] = ref_pair_costs(a_df, r_df, [tuple(p) for p in pairs[sel]], list(commonCT), dist_ct_coeff)
    out['costs_w2p5'] = ref_pair_costs(a_df, r_df, [tuple(p) for p in pairs[sel[:200]]], list(commonCT), 2.5)

    # a6 (scipy, host input) + a7 ------------------------------------------------------
    axy = a_df[['X', 'Y']].values
    tris0 = Delaunay(axy).simplices
    out['delaunay'] = tris0.astype(np.int64)
    for tag, kw in (('plain', dict(ignore_same_type_triangles=False, min_angle_deg=min_angle_deg)),
                    ('type', dict(ignore_same_type_triangles=True, min_angle_deg=min_angle_deg)),
                    ('noangle', dict(ignore_same_type_triangles=True, min_angle_deg=None)),
                    ('a30', dict(ignore_same_type_triangles=True, min_angle_deg=30, ensure_min_triangle_per_node=False))):
        kept, unc = quiet(ref.helpers.filter_triangles_by_radius, axy, tris0, radius, aligned_df=a_df,
                          remove_unconstrained_nodes=True, **kw)
        out[f'tri_{tag}'] = np.array(kept, dtype=np.int64).reshape(-1, 3)
        out[f'unc_{tag}'] = np.array(sorted(unc), dtype=np.int64)
        print(f"  filter[{tag}]: {len(tris0)} -> {len(kept)} (unconstrained {len(unc)})")
    # a tighter radius to exercise the radius rule and unconstrained nodes
    kept, unc = quiet(ref.helpers.filter_triangles_by_radius, axy, tris0, radius * 0.35, aligned_df=a_df,
                      ignore_same_type_triangles=True, remove_unconstrained_nodes=True, min_angle_deg=min_angle_deg)
    out['tri_tight'] = np.array(kept, dtype=np.int64).reshape(-1, 3)
    out['unc_tight'] = np.array(sorted(unc), dtype=np.int64)

    tris = out['tri_plain']
    # a8 -------------------------------------------------------------------------------
    w, s = ref_weights_signs(a_df, tris)
    out['tri_weights'], out['source_signs'] = w, s

    # a9 -------------------------------------------------------------------------------
    smap = simplex_map(len(a_df), tris)
    tinfo = ref.helpers.precompute_triangle_info(a_df, tris, smap)
    keys = list(tinfo.keys())
    out['tinfo_keys'] = np.array(keys, dtype=np.int64)
    out['tinfo_bounds'] = np.array([[tinfo[k]['bounds'][q] for q in ('min_x', 'max_x', 'min_y', 'max_y')] for k in keys]).reshape(-1, 4)
    out['tinfo_extreme'] = np.array([[tinfo[k][q] for q in ('max_x_vertex', 'min_x_vertex', 'max_y_vertex', 'min_y_vertex')] for k in keys],
                                    dtype=np.int64).reshape(-1, 4)

    # a5 -------------------------------------------------------------------------------
    full_costs = out['costs'] if len(sel) == len(pairs) else None
    if full_costs is None:
        # the start heuristics need every pair cost: take them from a vectorised evaluation that the
        # sampled literal loop above pins (checked equal on the sample before use)
        A = a_df[list(commonCT)].to_numpy(); R = r_df[list(commonCT)].to_numpy()
        d = np.abs(A[pairs[:, 0]] - R[pairs[:, 1]])
        acc = np.zeros(len(pairs))
        for t in range(d.shape[1]):
            acc = acc + d[:, t]
        rxy = r_df[['X', 'Y']].values
        full_costs = dist_ct_coeff * acc + (dist_ct_coeff * 0.001) * (np.abs(axy[pairs[:, 0], 0] - rxy[pairs[:, 1], 0]) +
                                                                      np.abs(axy[pairs[:, 0], 1] - rxy[pairs[:, 1], 1]))
        assert np.array_equal(full_costs[sel], out['costs'])
    out['all_costs'] = full_costs
    vp = [tuple(int(q) for q in p) for p in pairs]
    sizes = a_df['size'].to_numpy(dtype=float)
    chosen, unmatched = quiet(ref.init_helpers.compute_mip_start_pairs, valid_pairs=vp, costs=list(full_costs),
                              n_aligned=len(a_df), n_ref=len(r_df), aligned_sizes=sizes, no_match_penalty=no_match_penalty,
                              max_matches=1, init_method='greedy', verbose=False)
    out['greedy_chosen'] = np.array(chosen, dtype=np.int64).reshape(-1, 3)
    out['greedy_unmatched'] = np.array(sorted(unmatched), dtype=np.int64)
    # a low penalty makes "prefer unmatched" bite
    chosen_lo, unmatched_lo = quiet(ref.init_helpers.compute_mip_start_pairs, valid_pairs=vp, costs=list(full_costs),
                                    n_aligned=len(a_df), n_ref=len(r_df), aligned_sizes=sizes,
                                    no_match_penalty=float(np.median(full_costs)), max_matches=1, init_method='greedy', verbose=False)
    out['greedy_lo_penalty'] = np.array([float(np.median(full_costs))])
    out['greedy_lo_chosen'] = np.array(chosen_lo, dtype=np.int64).reshape(-1, 3)
    out['greedy_lo_unmatched'] = np.array(sorted(unmatched_lo), dtype=np.int64)
    if do_hungarian:
        ch, un = quiet(ref.init_helpers.compute_mip_start_pairs, valid_pairs=vp, costs=list(full_costs),
                       n_aligned=len(a_df), n_ref=len(r_df), aligned_sizes=sizes, no_match_penalty=no_match_penalty,
                       max_matches=1, init_method='hungarian', init_hungarian_max_n=100000, verbose=False)
        out['hungarian_chosen'] = np.array(ch, dtype=np.int64).reshape(-1, 3)
        out['hungarian_unmatched'] = np.array(sorted(un), dtype=np.int64)
    print(f"  greedy start: {len(chosen)} matched / {len(unmatched)} unmatched")

    # a10 under the greedy matching ------------------------------------------------------
    x_vals = np.zeros(len(pairs))
    x_vals[out['greedy_chosen'][:, 2]] = 1.0
    out['x_vals'] = x_vals
    checked, viol = ref_lazy_sweep(x_vals, vp, tris, s, r_df)
    out['lazy_checked'] = np.array([checked], dtype=np.int64)
    out['lazy_violating'] = viol
    print(f"  lazy sweep: checked={checked} violating={len(viol)}")

    # a11 --------------------------------------------------------------------------------
    matches_df = pd.DataFrame({'aligned_idx': out['greedy_chosen'][:, 0], 'ref_idx': out['greedy_chosen'][:, 1]})
    v = ref.violationhelper.verify_spatial_preservation(aligned_df=a_df, ref_df=r_df, matches_df=matches_df, triangle_info=tinfo)
    out.update(violations_to_arrays(v))
    print(f"  xy-order: {v['violation_summary']}")

    # a12 --------------------------------------------------------------------------------
    a2r = {int(i): int(j) for i, j, _ in out['greedy_chosen']}
    before, after, flipped, m3 = ref_area_flips(a_df, r_df, tris, a2r)
    out.update({'area_before': before, 'area_after': after, 'area_flipped': flipped, 'area_matched3': m3})
    print(f"  area flips: {len(flipped)}")

    # a14 (sample) -----------------------------------------------------------------------
    rng = np.random.default_rng(seed + 7)
    rdict = {i: {'X': r_df.iloc[i]['X'], 'Y': r_df.iloc[i]['Y']} for i in range(len(r_df))}
    combos = rng.integers(0, len(r_df), size=(2000, 3))
    out['eager_combos'] = combos.astype(np.int64)
    out['eager_signs'] = np.array([ref.helpers.calc_ref_area(((int(a), int(b), int(c)), rdict)) for a, b, c in combos], dtype=np.int8)

    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **out)


def add_row_ids(df):
    df = df.copy()
    df['__row'] = np.arange(len(df))
    if 'size' not in df.columns:
        df['size'] = 1
    return df


# ------------------------------------------------------------------ restricted unpickler
class _Restricted(pickle.Unpickler):
    ALLOWED = {('numpy.core.multiarray', '_reconstruct'), ('numpy._core.multiarray', '_reconstruct'),
               ('numpy', 'ndarray'), ('numpy', 'dtype'), ('numpy.core.multiarray', 'scalar'),
               ('numpy._core.multiarray', 'scalar'), ('builtins', 'set')}

    def find_class(self, module, name):
        if (module, name) in self.ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"blocked global {module}.{name}")


def load_var_out(path):
    with open(path, 'rb') as f:
        version = np.lib.format.read_magic(f)
        np.lib.format._read_array_header(f, version)
        obj = _Restricted(f).load()
    return obj.item() if isinstance(obj, np.ndarray) else obj


def stored_run_case(name):
    """examples/simulated_{st,elastic}: outputs stored by the reference authors' own run."""
    d = os.path.join(REF_ROOT, 'examples', name)
    a_df = pd.read_csv(os.path.join(d, 'aligned_df.csv'))
    r_df = pd.read_csv(os.path.join(d, 'ref_df.csv'))
    m_df = pd.read_csv(os.path.join(d, 'matches_df.csv'))
    var_out = load_var_out(os.path.join(d, 'var_out.npy'))
    tinfo = var_out['triangle_data']['triangle_info']
    keys = list(tinfo.keys())
    out = {'aligned_xy': a_df[['X', 'Y']].to_numpy(dtype=np.float64), 'ref_xy': r_df[['X', 'Y']].to_numpy(dtype=np.float64),
           'matches': m_df[['aligned_idx', 'ref_idx']].to_numpy(dtype=np.int64),
           'tinfo_keys': np.array(keys, dtype=np.int64),
           'tinfo_vertices': np.array([list(tinfo[k]['vertices']) for k in keys], dtype=np.int64)}
    stored = violations_to_arrays(var_out['violations'])
    out.update({f'stored_{k}': v for k, v in stored.items()})
    v = ref.violationhelper.verify_spatial_preservation(aligned_df=a_df, ref_df=r_df, matches_df=m_df, triangle_info=tinfo)
    out.update(violations_to_arrays(v))
    print(f"[{name}] stored summary {stored['viol_summary']} recomputed {out['viol_summary']}")
    np.savez_compressed(os.path.join(OUT, f'{name}.npz'), **out)


def adversarial_case():
    """Exact ties, collinear triples, duplicates, unmatched vertices, empty inputs."""
    out = {}
    # regular grid: many exact distance ties; reference order inside a tie group is unspecified
    g = np.stack(np.meshgrid(np.arange(8.0), np.arange(8.0)), -1).reshape(-1, 2)
    a_df = add_row_ids(pd.DataFrame({'X': g[:, 0] + 0.25, 'Y': g[:, 1] + 0.25, 'cell_type': 'c1', 'c1': 100.0}))
    r_df = add_row_ids(pd.DataFrame({'X': g[:, 0], 'Y': g[:, 1], 'cell_type': 'c1', 'c1': 100.0}))
    _, _, pairs = quiet(ref.utils.find_knn_within_radius, a_df, r_df, 1.5, knn=6)
    out['grid_axy'], out['grid_rxy'], out['grid_pairs'] = a_df[['X', 'Y']].values, r_df[['X', 'Y']].values, np.asarray(pairs, dtype=np.int64)
    # sparse coverage: small radius drops aligned rows and refs, so the compaction re-indexing matters
    rs = synth.make_cells(400, 4, seed=3, side=100.0)
    ms = synth.make_cells(300, 4, seed=4, side=100.0)
    sa, sr = add_row_ids(synth.to_frame(ms)), add_row_ids(synth.to_frame(rs))
    na, nr, pairs = quiet(ref.utils.find_knn_within_radius, sa, sr, 4.0, knn=3)
    out['sparse_axy'], out['sparse_rxy'] = ms['xy'], rs['xy']
    out['sparse_pairs'] = np.asarray(pairs, dtype=np.int64)
    out['sparse_kept_aligned'] = na['__row'].to_numpy(dtype=np.int64)
    out['sparse_kept_ref'] = nr['__row'].to_numpy(dtype=np.int64)
    # radius boundary: points at exactly r (3-4-5 triangles) must be included (<=)
    a2 = add_row_ids(pd.DataFrame({'X': [0.0, 10.0], 'Y': [0.0, 10.0], 'cell_type': 'c1', 'c1': 1.0}))
    r2 = add_row_ids(pd.DataFrame({'X': [3.0, 4.0, 5.0, 13.0, 10.0, 3.0000000001], 'Y': [4.0, 3.0, 0.0, 14.0, 15.0000001, 4.0],
                                   'cell_type': 'c1', 'c1': 1.0}))
    na, nr, pairs = quiet(ref.utils.find_knn_within_radius, a2, r2, 5.0, knn=4)
    out['edge_axy'], out['edge_rxy'] = a2[['X', 'Y']].values, r2[['X', 'Y']].values
    out['edge_pairs'] = np.asarray(pairs, dtype=np.int64)
    out['edge_kept_ref'] = nr['__row'].to_numpy(dtype=np.int64)
    # triangles: collinear, duplicate vertex, right isosceles at the 45-degree threshold, tiny
    pts = np.array([[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [5, 5], [5.0000001, 5], [5, 5.0000001], [0, 0], [3, 0], [0, 3]], dtype=float)
    tr = np.array([[0, 1, 2], [0, 1, 3], [1, 4, 3], [5, 6, 7], [0, 8, 1], [0, 9, 10], [3, 4, 1], [2, 1, 0]])
    tdf = add_row_ids(pd.DataFrame({'X': pts[:, 0], 'Y': pts[:, 1], 'cell_type': ['a', 'a', 'a', 'b', 'a', 'a', 'a', 'a', 'a', 'b', 'b']}))
    out['adv_pts'], out['adv_tris'] = pts, tr.astype(np.int64)
    out['adv_type'] = tdf['cell_type'].to_numpy().astype(str)
    for tag, kw in (('45', dict(min_angle_deg=45, ignore_same_type_triangles=False)),
                    ('45t', dict(min_angle_deg=45, ignore_same_type_triangles=True)),
                    ('none', dict(min_angle_deg=None, ignore_same_type_triangles=True)),
                    ('0', dict(min_angle_deg=0, ignore_same_type_triangles=False)),
                    ('15', dict(min_angle_deg=15, ignore_same_type_triangles=True))):
        for rad in (10.0, 3.0, 1.0):
            with np.errstate(all='ignore'):
                kept, unc = quiet(ref.helpers.filter_triangles_by_radius, pts, tr, rad, aligned_df=tdf, remove_unconstrained_nodes=True, **kw)
            out[f'adv_kept_{tag}_{rad}'] = np.array(kept, dtype=np.int64).reshape(-1, 3)
            out[f'adv_unc_{tag}_{rad}'] = np.array(sorted(unc), dtype=np.int64)
    w, s = ref_weights_signs(add_row_ids(tdf), tr)
    out['adv_signs'] = s
    # empty triangle list
    kept = quiet(ref.helpers.filter_triangles_by_radius, pts, np.zeros((0, 3), dtype=int), 10.0, aligned_df=tdf, ignore_same_type_triangles=True)
    out['adv_empty_kept'] = np.array(kept, dtype=np.int64).reshape(-1, 3)
    # sweeps with unmatched vertices, duplicate aligned rows in matches (last wins), collinear ref triple
    rxy = np.array([[0, 0], [1, 0], [2, 0], [0, 1], [1, 1], [0.5, 0.5], [2, 2]], dtype=float)
    rdf = pd.DataFrame({'X': rxy[:, 0], 'Y': rxy[:, 1]})
    adf = pd.DataFrame({'X': pts[:, 0], 'Y': pts[:, 1]})
    m_df = pd.DataFrame({'aligned_idx': [0, 1, 3, 4, 1, 9, 10], 'ref_idx': [4, 1, 0, 3, 2, 5, 6]})
    smap = simplex_map(len(adf), tr)
    tinfo = ref.helpers.precompute_triangle_info(adf, tr, smap)
    v = ref.violationhelper.verify_spatial_preservation(aligned_df=adf, ref_df=rdf, matches_df=m_df, triangle_info=tinfo)
    out['sw_rxy'] = rxy
    out['sw_matches'] = m_df.to_numpy(dtype=np.int64)
    out['sw_tinfo_keys'] = np.array(list(tinfo.keys()), dtype=np.int64)
    out.update({f'sw_{k}': val for k, val in violations_to_arrays(v).items()})
    a2r = {}
    for a, r in m_df.to_numpy():
        a2r[int(a)] = int(r)
    before, after, flipped, m3 = ref_area_flips(adf, rdf, tr, a2r)
    out.update({'sw_area_before': before, 'sw_area_after': after, 'sw_area_flipped': flipped, 'sw_area_matched3': m3})
    # lazy sweep: pairs listed so that a later x>0.5 overrides an earlier one
    vp = [(0, 4), (1, 1), (1, 2), (3, 0), (4, 3), (9, 5), (10, 6), (2, 2)]
    xv = np.array([1, 1, 0.9, 1, 0.6, 1, 1, 0.2])
    checked, viol = ref_lazy_sweep(xv, vp, tr, s, rdf)
    out['sw_pairs'], out['sw_x'] = np.array(vp, dtype=np.int64), xv
    out['sw_lazy_checked'], out['sw_lazy_violating'] = np.array([checked]), viol
    np.savez_compressed(os.path.join(OUT, 'adversarial.npz'), **out)
    print(f"[adversarial] grid pairs {len(out['grid_pairs'])}, edge pairs {out['edge_pairs'].tolist()}, lazy {checked}/{viol.tolist()}")


def eval_case():
    """eval_utils.check_triangle_violations (src/eval_utils.py:66-223) on the shipped synthetic example."""
    d = os.path.join(REF_ROOT, 'examples', 'synthetic', 'data')
    ref_df = add_row_ids(pd.read_csv(os.path.join(d, 'ref.csv'), index_col=0))
    qry_df = add_row_ids(pd.read_csv(os.path.join(d, 'query.csv'), index_col=0))
    a_df, r_df, pairs = quiet(ref.utils.find_knn_within_radius, qry_df, ref_df, 5, knn=8)
    pairs = np.asarray(pairs)
    rng = np.random.default_rng(11)
    first = np.flatnonzero(np.r_[True, pairs[1:, 0] != pairs[:-1, 0]])
    pick = first + np.minimum(rng.integers(0, 3, len(first)), np.diff(np.r_[first, len(pairs)]) - 1)  # one of the 3 nearest
    sel = pairs[pick]
    sel = sel[rng.random(len(sel)) > 0.1]                      # ~10 % unmatched
    ids = 1000 + 3 * np.arange(len(a_df))                      # metacell ids are labels, not positions
    mdf = pd.DataFrame({'X': a_df['X'].to_numpy(), 'Y': a_df['Y'].to_numpy()}, index=ids)
    tri_ids = ids[Delaunay(mdf[['X', 'Y']].to_numpy()).simplices]
    tri_ids = np.vstack([tri_ids, [[ids[0], ids[1], 999999]]])  # a triangle with an id unknown to metacell_df
    out_df = pd.DataFrame({'aligned_metacell_index': ids[sel[:, 0]], 'matched_ref_index': sel[:, 1],
                           'mapped_x': r_df['X'].to_numpy()[sel[:, 1]], 'mapped_y': r_df['Y'].to_numpy()[sel[:, 1]],
                           'cell_type': a_df['cell_type'].to_numpy()[sel[:, 0]]})
    out_df = pd.concat([out_df, out_df.iloc[[5]].assign(mapped_x=out_df['mapped_x'].iloc[7], mapped_y=out_df['mapped_y'].iloc[9])],
                       ignore_index=True)                       # duplicate id: the last row wins
    out_df.loc[20, 'mapped_x'] = np.nan                         # NaN sign -> counted as a flip by the reference

    class MC:
        metacell_df = mdf
        metacell_delaunay = tri_ids

    out = {'ids': ids, 'mxy': mdf[['X', 'Y']].to_numpy(), 'tri_ids': tri_ids.astype(np.int64),
           'o_id': out_df['aligned_metacell_index'].to_numpy(), 'o_ref': out_df['matched_ref_index'].to_numpy(),
           'o_mx': out_df['mapped_x'].to_numpy(), 'o_my': out_df['mapped_y'].to_numpy(),
           'o_type': out_df['cell_type'].to_numpy().astype(str)}
    keys = ('total_triangles', 'triangles_with_all_matched', 'triangles_processed', 'triangles_same_type_skipped',
            'triangles_flipped', 'percent_flipped', 'nodes_in_violating_triangles', 'percent_nodes_violating')
    for tag, kw in (('default', {}), ('alltypes', dict(ignore_same_type_triangles=False)),
                    ('local', dict(node_local=True)), ('local_strict', dict(node_local=True, majority_threshold=0.3, min_flips=2)),
                    ('local_alltypes', dict(node_local=True, ignore_same_type_triangles=False, majority_threshold=0.75))):
        df, stats = ref.eval_utils.check_triangle_violations(out_df, MC(), **kw)
        out[f'viol_{tag}'] = df['in_violating_triangle'].to_numpy().astype(np.uint8)
        out[f'stats_{tag}'] = np.array([stats[k] for k in keys], dtype=np.float64)
        print(f"[eval_tri/{tag}] {stats}")
    # check_alignment (src/eval_utils.py:6-55) on tie-free coordinates
    q = synth.to_frame(synth.make_cells(700, 4, seed=31, side=100.0))
    t = synth.to_frame(synth.make_cells(900, 4, seed=32, side=100.0))
    for k in (1, 5):
        df, score = ref.eval_utils.check_alignment(q, t, 'X', 'Y', kNN=k)
        out[f'align_match_{k}'] = df[f'_{k}NN_match'].to_numpy().astype(np.uint8)
        out[f'align_score_{k}'] = np.array([score])
        if k == 1:
            out['align_ctype_1'] = df['_1NN_match_ctype'].to_numpy().astype(str)
    np.savez_compressed(os.path.join(OUT, 'eval_tri.npz'), **out)


def metacell_arrays(prefix, mdf, tri, orig_tri, num_cols, other_cols):
    members = mdf['members'].tolist()
    return {f'{prefix}_xy': mdf[['X', 'Y']].to_numpy(dtype=np.float64), f'{prefix}_size': mdf['size'].to_numpy(dtype=np.int64),
            f'{prefix}_type': mdf['cell_type'].to_numpy().astype(str),
            f'{prefix}_members': np.array([m for ms in members for m in ms], dtype=np.int64),
            f'{prefix}_member_off': np.cumsum([0] + [len(ms) for ms in members]).astype(np.int64),
            f'{prefix}_num': mdf[num_cols].to_numpy(dtype=np.float64), f'{prefix}_other': mdf[other_cols].to_numpy().astype(str),
            f'{prefix}_mcid': mdf['metacell_id'].to_numpy(dtype=np.int64), f'{prefix}_cols': np.array(list(mdf.columns)),
            f'{prefix}_tri': np.asarray(tri, dtype=np.int64).reshape(-1, 3), f'{prefix}_orig_tri': np.asarray(orig_tri, dtype=np.int64).reshape(-1, 3)}


def metacell_case():
    """metacell_utils.greedy_triangle_collapse (src/metacell_utils.py:160-561) on the shipped example and a seeded case."""
    out = {}
    d = os.path.join(REF_ROOT, 'examples', 'synthetic', 'data')
    q = pd.read_csv(os.path.join(d, 'query.csv'), index_col=0)
    np.savez_compressed(os.path.join(OUT, 'metacell_inputs.npz'), q_xy=q[['X', 'Y']].to_numpy(), q_type=q['cell_type'].to_numpy().astype(str),
                        q_c=q[['c1', 'c2', 'c3']].to_numpy(), q_quadrant=q['quadrant'].to_numpy().astype(str), q_idx=q['cell_idx'].to_numpy())
    q = q[['X', 'Y', 'cell_type', 'c1', 'c2', 'c3', 'quadrant', 'cell_idx']]
    for tag, kw in (('s3', dict(max_metacell_size=3, r_max=5, min_angle_deg=5)),
                    ('s6', dict(max_metacell_size=6, r_max=None, min_angle_deg=10)),
                    ('s9', dict(max_metacell_size=9, r_max=4, min_angle_deg=None)),
                    ('s1', dict(max_metacell_size=1, r_max=5, min_angle_deg=5))):
        mc = quiet(ref.metacell_utils.greedy_triangle_collapse, q, original_idx_col='cell_idx', use_alpha_shape=False,
                   return_object=True, **kw)
        out.update(metacell_arrays(f'q_{tag}', mc.metacell_df, mc.metacell_delaunay, mc.original_delaunay, ['c1', 'c2', 'c3'], ['quadrant']))
        print(f"[metacell/query {tag}] {len(q)} -> {len(mc.metacell_df)} metacells, {len(mc.metacell_delaunay)} triangles, max size {mc.metacell_df['size'].max()}")
    cells = synth.make_cells(1500, 4, seed=5)
    df = synth.to_frame(cells)
    df['batch'] = np.where(np.arange(len(df)) % 3 == 0, 'b0', 'b1')
    df['flag'] = (np.arange(len(df)) % 2).astype(np.int64)
    mdf, tri = quiet(ref.metacell_utils.greedy_triangle_collapse, df, max_metacell_size=5, r_max=30, min_angle_deg=12)
    out['seeded_n'] = np.array([len(df)])
    out.update(metacell_arrays('seeded', mdf, tri, np.zeros((0, 3)), ['c1', 'c2', 'c3', 'c4', 'size', 'flag'], ['batch']))
    print(f"[metacell/seeded] {len(df)} -> {len(mdf)} metacells, {len(tri)} triangles")
    np.savez_compressed(os.path.join(OUT, 'metacell.npz'), **out)


def merge_dedup_case():
    """merge_window_matches_unique_ref (src/helpers.py:692-815) on tables that show WHICH duplicate its de-duplication keeps
    (:745-753): every (aligned, ref) pair is disjoint from every other, so the matching keeps them all, and each is proposed
    by 1-6 windows with varying violation flags (some missing) and window ids, ties included; X carries the input row number,
    so the output names the surviving row of every pair."""
    rng = np.random.default_rng(77)
    out = {}
    for tag, n_pairs, n_win in (('a', 800, 16), ('b', 60, 3)):
        reps = rng.integers(1, 7, n_pairs)
        a = np.repeat(np.arange(n_pairs) * 3 + 11, reps)
        perm = rng.permutation(len(a))
        a = a[perm]
        dfm = pd.DataFrame({'window_id': rng.integers(0, n_win, len(a)), 'Aligned_Cell_Num_Old': a, 'Ref_Cell_Num_Old': 5000 + 2 * a,
                            'X': np.arange(len(a), dtype=float), 'Y': 0.0,
                            'filtered_violation': (rng.random(len(a)) < 0.4).astype(object)})
        dfm.loc[rng.choice(len(a), len(a) // 25, replace=False), 'filtered_violation'] = np.nan
        cuts = np.sort(rng.choice(np.arange(1, len(a)), 4, replace=False))
        lst = [part.copy() for part in np.split(dfm, cuts)]        # the caller hands over several tables; order = concatenation order
        res = ref.helpers.merge_window_matches_unique_ref(lst)
        out[f'{tag}_in'] = np.column_stack((dfm['window_id'], dfm['Aligned_Cell_Num_Old'], dfm['Ref_Cell_Num_Old'])).astype(np.int64)
        out[f'{tag}_in_viol'] = dfm['filtered_violation'].astype(float).to_numpy()          # NaN = missing
        out[f'{tag}_cuts'] = cuts.astype(np.int64)
        out[f'{tag}_out_rows'] = res['X'].to_numpy().astype(np.int64)                      # input row of every kept pair, output order
        out[f'{tag}_out'] = res[['window_id', 'Aligned_Cell_Num_Old', 'Ref_Cell_Num_Old']].to_numpy(dtype=np.int64)
        out[f'{tag}_out_viol'] = res['filtered_violation'].to_numpy().astype(np.uint8)
        print(f"[merge/dedup {tag}] {len(dfm)} rows, {n_pairs} pairs -> {len(res)} kept")
    np.savez_compressed(os.path.join(OUT, 'merge_dedup.npz'), **out)


def unpack_merge_case():
    """unpack_metacell_matches (src/metacell_utils.py:564-766) and merge_window_matches_unique_ref (src/helpers.py:692-815)."""
    out = {}
    cells = synth.make_cells(900, 3, seed=21)
    a_df = synth.to_frame(cells); a_df['Cell_Num_Old'] = np.arange(len(a_df)) * 2 + 5
    r_df = synth.to_frame(synth.make_jittered(cells, seed=22)); r_df['Cell_Num_Old'] = np.arange(len(r_df)) * 3 + 1
    mca = quiet(ref.metacell_utils.greedy_triangle_collapse, a_df, max_metacell_size=5, r_max=40, min_angle_deg=10, return_object=True)
    mcr = quiet(ref.metacell_utils.greedy_triangle_collapse, r_df, max_metacell_size=4, r_max=40, min_angle_deg=10, return_object=True)
    rng = np.random.default_rng(4)
    n = 400
    mm = pd.DataFrame({'Aligned_metacell_id': rng.choice(len(mca.metacell_df), n, replace=False),
                       'Ref_metacell_id': rng.choice(len(mcr.metacell_df), n, replace=False)})
    out['mm'] = mm.to_numpy(dtype=np.int64)
    for tag, kw in (('dist_both', dict(strategy='distribute')),
                    ('near_both', dict(strategy='nearest', aligned_df=a_df, ref_df=r_df, aligned_original_idx_col='Cell_Num_Old',
                                       ref_original_idx_col='Cell_Num_Old'))):
        res = ref.metacell_utils.unpack_metacell_matches(mm, mca.metacell_df, mcr.metacell_df, **kw)
        out[f'unpack_{tag}'] = res[['Aligned_cell_id', 'Ref_cell_id']].to_numpy(dtype=np.int64)
    res = ref.metacell_utils.unpack_metacell_matches(mm.assign(Ref_metacell_id=mm['Ref_metacell_id'] % len(r_df)), mca.metacell_df, r_df)
    out['unpack_simple'] = res[['Aligned_cell_id', 'Ref_cell_id']].to_numpy(dtype=np.int64)
    # window merge: overlapping windows proposing conflicting matches; built so that the maximum matching is unique
    rows = []
    for w in range(3):
        for q in range(40):
            a = 10 * w + q
            rows.append({'window_id': w, 'Aligned_Cell_Num_Old': a, 'Ref_Cell_Num_Old': 1000 + a, 'X': float(a), 'Y': float(w),
                         'filtered_violation': bool((a + w) % 5 == 0)})
    dfm = pd.DataFrame(rows)
    dfm.loc[dfm.index[7], 'filtered_violation'] = np.nan
    lst = [dfm[dfm.window_id == w].copy() for w in range(3)]
    res = ref.helpers.merge_window_matches_unique_ref(lst)
    out['merge_rows'] = res[['window_id', 'Aligned_Cell_Num_Old', 'Ref_Cell_Num_Old']].to_numpy(dtype=np.int64)
    out['merge_viol'] = res['filtered_violation'].to_numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, 'unpack_merge.npz'), **out)
    print(f"[unpack/merge] {[ (k, v.shape) for k, v in out.items()]}")


def eager_model_case():
    """lazy_constraints=False: add_spatial_constraints_triangle_based (src/helpers.py:444-573) RUN AS-IS against a recording
    stand-in for the solver API (tests/fake_gurobipy.py); the fixture holds the variables and constraints it emitted."""
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import fake_gurobipy as fg

    radius, knn = 14.0, 3
    cells = synth.make_cells(140, 3, seed=31)
    r_df = synth.to_frame(cells)
    r_df.loc[r_df.index % 7 == 0, 'size'] = 3.0          # some reference metacells
    a_df = synth.to_frame(synth.make_jittered(cells, seed=32))
    new_a, new_r, pairs = quiet(ref.utils.find_knn_within_radius, a_df, r_df, radius, knn)
    valid_pairs = pairs
    coords = new_a[['X', 'Y']].values
    tris = Delaunay(coords).simplices
    kept = quiet(ref.helpers.filter_triangles_by_radius, coords, tris, radius, aligned_df=new_a,
                 ignore_same_type_triangles=True, min_angle_deg=15)
    _, _, valid_pairs_map = ref.helpers.precompute_coordinate_maps(new_a, new_r, valid_pairs)
    simplex_map = {i: set() for i in range(len(new_a))}
    for idx, simplex in enumerate(kept):
        for i in simplex:
            simplex_map[i].add(idx)
    model = fg.Model('optimal_matches')
    x = model.addVars(len(valid_pairs), vtype=fg.GRB.BINARY, lb=0, ub=1, name='x')
    penalty_vars = model.addVars(len(new_r), vtype=fg.GRB.CONTINUOUS, lb=0, ub=1000, name='penalty')   # src/same.py:1116-1118
    no_match_vars = model.addVars(len(new_a), vtype=fg.GRB.CONTINUOUS, lb=0, ub=1, name='no_match')
    n_x = len(model.vars)
    ref.helpers.GRB = fg.GRB
    ref.helpers.quicksum = fg.quicksum
    # assignment constraints, run as-is (src/helpers.py:102-161); some ref rows are metacells so both limits occur
    quiet(ref.helpers.add_basic_constraints_optimized, model, valid_pairs, len(new_r), len(new_a), 1, x, penalty_vars, no_match_vars,
          aligned_df=new_a, ref_df=new_r, ref_metacell_match_multiplier=None)
    apv, zpv = quiet(ref.helpers.add_spatial_constraints_triangle_based, model, valid_pairs_map, x, new_a, new_r, valid_pairs,
                     simplex_map, kept)
    names = [v.VarName for v in model.vars]
    index = {n: i for i, n in enumerate(names)}
    canon = fg.canonical_constraints(model.constrs)
    sense = np.array([{'<=': -1, '==': 0, '>=': 1}[c[0]] for c in canon], dtype=np.int8)
    const = np.array([c[1] for c in canon])
    width = max(len(c[2]) for c in canon)
    term_var = np.full((len(canon), width), -1, np.int32)
    term_coef = np.zeros((len(canon), width))
    for q, c in enumerate(canon):
        for t, (n, k) in enumerate(c[2]):
            term_var[q, t], term_coef[q, t] = index[n], k
    out = {'params': np.array([radius, knn]), 'n_x': np.array([n_x]), 'pairs': np.asarray(valid_pairs, dtype=np.int64),
           'triangles': np.asarray(kept, dtype=np.int64), 'var_names': np.array(names), 'var_lb': np.array([v.lb for v in model.vars], dtype=float),
           'var_ub': np.array([np.inf if v.ub is None else v.ub for v in model.vars], dtype=float),
           'area_penalty_names': np.array([v.VarName for v in apv]), 'z_names': np.array([v.VarName for v in zpv]),
           'constr_names': np.array([n or '' for n, _ in model.constrs]),
           'sense': sense, 'const': const, 'term_var': term_var, 'term_coef': term_coef}
    zero = int(((term_coef == 0).all(axis=1) | ((term_var >= 0).sum(axis=1) == 1)).sum())
    print(f"[eager model] {len(kept)} triangles, {len(zpv)} z vars, {len(canon)} constraints, {zero} with a zero orientation product")
    np.savez_compressed(os.path.join(OUT, 'eager_model.npz'), **out)


def run_same_mock_case():
    """run_same (src/same.py:706-1489) and sliding_window_matching (:297-590) RUN AS-IS with the recording solver double in
    place of gurobipy: `optimize` takes the MIP start as the incumbent, calls the lazy callback once and reports OPTIMAL.
    Pins everything on either side of the solve: model variables / constraints / objective / starts / parameters, the
    cuts the callback adds, the match table, var_out, the files written, the window tiling and central trimming."""
    import fake_gurobipy as fg
    import run_same_record as rec
    import shutil
    import tempfile

    out = {}
    work = tempfile.mkdtemp(prefix='same_golden_')
    cwd = os.getcwd()
    os.chdir(work)                                           # run_same writes gurobi_logs/ and matching_model.lp into the cwd
    try:
        cells = synth.make_cells(300, 4, seed=41)
        r_df = synth.to_frame(cells)
        a_df = synth.to_frame(synth.make_jittered(cells, seed=42))
        cols = synth.type_columns(4)
        cases = {
            'lazy_greedy': (dict(radius=20, knn=4), dict(init_method='greedy', lazy_allowed_flip_fraction=0.0, lazy_max_cuts_per_incumbent=25), {}),
            'priority_hungarian': (dict(radius=20, knn=4, ignore_knn_if_matched=True, min_angle_deg=None, ignore_same_type_triangles=False,
                                        dist_ct_coeff=2.5, no_match_penalty=40, penalty_coeff=3.0, delaunay_penalty=7.0),
                                   dict(init_method='hungarian', lazy_allowed_flip_fraction=0.0, lazy_max_cuts=9, time_limit=60, mip_focus=1,
                                        cuts=2, heuristics=0.2), {}),
            'eager': (dict(radius=14, knn=3, lazy_constraints=False), dict(init_method='greedy'), {}),
            # max_matches=2: no MIP start is applied (src/init_helpers.py:84-92); the callback sees the all-zero incumbent
            'max_matches2': (dict(radius=20, knn=5, max_matches=2, min_angle_deg=0), dict(init_method='greedy', lazy_allowed_flip_fraction=0.0), {}),
            # reference metacells with an explicit multiplier; Hungarian start skipped because the problem is above init_hungarian_max_n
            'multiplier': (dict(radius=25, knn=6, ref_metacell_match_multiplier=2), dict(init_method='hungarian', init_hungarian_max_n=100,
                                                                                       lazy_allowed_flip_fraction=0.0), {}),
        }
        # caller-supplied triangulation in vertex-id space, missing some nodes -> unconstrained nodes are removed (:1054-1083)
        ids = np.arange(len(a_df)) * 5 + 2
        a_pre = a_df.assign(mc_id=ids)
        tri_all = Delaunay(a_pre[['X', 'Y']].values).simplices
        keep = ~np.isin(tri_all, np.arange(0, len(a_pre), 9)).any(axis=1)
        cases['precomputed'] = (dict(radius=20, knn=4), dict(init_method='greedy', lazy_allowed_flip_fraction=0.0),
                                dict(aligned_delaunay=ids[tri_all[keep]], aligned_delaunay_vertex_col='mc_id'))
        r_meta = r_df.copy()
        r_meta.loc[r_meta.index % 5 == 0, 'size'] = 4.0
        for tag, (op, gpar, extra) in cases.items():
            adf = a_pre if tag == 'precomputed' else a_df
            outprefix = os.path.join(work, tag)
            out_df, var_out = quiet(ref.same.run_same, (r_meta if tag == 'multiplier' else r_df).copy(), adf.copy(), cols, outprefix=outprefix,
                                    optim_params=ref.same.init_optim_params(**op), gurobi_params=ref.same.init_gurobi_params(**gpar), **extra)
            r = rec.record_run(out_df, var_out, fg.Model.last)
            r['files'] = np.array(sorted(os.listdir(outprefix)), dtype=str)
            out.update({f'{tag}/{k}': v for k, v in r.items()})
            print(f"[run_same/{tag}] {len(out_df)} matches, {len(fg.Model.last.vars)} vars, {len(fg.Model.last.constrs)} constraints, "
                  f"{len(fg.Model.last.lazy)} cuts")

        # the metacell flow: collapse both sections, match metacells (MetaCell object passed as aligned_df: frame, triangulation,
        # vertex column and cell_id_col are taken from it, src/same.py:889-899), unpack to cells with per-match assignments
        cells = synth.make_cells(900, 3, seed=61)
        r_c = synth.to_frame(cells)
        a_c = synth.to_frame(synth.make_jittered(cells, seed=62))
        a_c['Cell_Num_Old'] = np.arange(len(a_c)) * 2 + 7
        mc_a = quiet(ref.metacell_utils.greedy_triangle_collapse, a_c, max_metacell_size=4, r_max=40, min_angle_deg=10, return_object=True)
        mc_r = quiet(ref.metacell_utils.greedy_triangle_collapse, r_c, max_metacell_size=3, r_max=40, min_angle_deg=10, return_object=True)
        # the MetaCell container's own helpers (src/metacell_utils.py:55-157) on the reference object
        probe = np.vstack([mc_a.original_delaunay[:5], [[7, 9, 123456789]], mc_a.original_delaunay[5:8]])   # one unknown vertex id
        out['mc_helpers/original_delaunay'] = np.asarray(mc_a.original_delaunay)
        out['mc_helpers/members_0_5_last'] = np.array([json.dumps([int(v) for v in mc_a.metacell_members(k)])
                                                       for k in (0, 5, len(mc_a.metacell_df) - 1)])
        out['mc_helpers/rows'] = mc_a.original_delaunay_to_row_indices()
        out['mc_helpers/rows_probe_drop'] = mc_a.original_delaunay_to_pos(probe)
        out['mc_helpers/xy'] = mc_a.original_delaunay_to_xy()
        out['mc_helpers/xy_probe'] = mc_a.original_delaunay_to_xy(probe, on_missing='drop')
        out['mc_helpers/mc_xy'] = mc_a.metacell_delaunay_to_xy()
        out['mc_helpers/summary'] = np.array([json.dumps(mc_a.to_summary_dict(), sort_keys=True, default=str)])
        try:
            mc_a.original_delaunay_to_row_indices(probe, on_missing='error')
            out['mc_helpers/error_raises'] = np.array([0])
        except KeyError:
            out['mc_helpers/error_raises'] = np.array([1])
        mop = dict(radius=30, knn=4)
        mgp = dict(init_method='greedy', lazy_allowed_flip_fraction=0.0, lazy_max_cuts_per_incumbent=40)
        out_df, var_out = quiet(ref.same.run_same, mc_r.metacell_df, mc_a, synth.type_columns(3), outprefix=os.path.join(work, 'mc'),
                                optim_params=dict(mop), gurobi_params=dict(mgp))
        r = rec.record_run(out_df.drop(columns=['members'], errors='ignore'), var_out, fg.Model.last)
        out.update({f'metacell_flow/{k}': v for k, v in r.items()})
        indiv = ref.metacell_utils.unpack_metacell_matches(out_df, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_c, ref_df=r_c,
                                                           strategy='nearest', aligned_original_idx_col='Cell_Num_Old',
                                                           ref_original_idx_col='Cell_Num_Old')
        out['metacell_flow_unpacked'] = indiv[['Aligned_cell_id', 'Ref_cell_id']].to_numpy(dtype=np.int64)
        print(f"[metacell flow] {len(mc_a.metacell_df)} x {len(mc_r.metacell_df)} metacells, {len(out_df)} metacell matches -> {len(indiv)} cell matches")
        res_mc = quiet(ref.same.sliding_window_matching, mc_r, mc_a, commonCT=synth.type_columns(3),
                       optim_params=dict(mop, window_size=200, overlap=50, min_cells_per_window=20), gurobi_params=dict(mgp))
        out.update({f'sw_metacell/{k}': v for k, v in rec.record_frame('res', res_mc).items()})
        print(f"[metacell windows] {len(res_mc)} central matches, columns {list(res_mc.columns)}")

        # the shipped example exactly as examples/synthetic/run_same.sh drives it (its parameter block :31-54 and calls :84-131):
        # size-1 "metacells" (Delaunay + filter only), MetaCell objects into sliding_window_matching, commonCT inferred
        d = os.path.join(REF_ROOT, 'examples', 'synthetic', 'data')
        ex_ref = pd.read_csv(os.path.join(d, 'ref.csv'), index_col=0)
        ex_query = pd.read_csv(os.path.join(d, 'query.csv'), index_col=0)
        out.update({f'example/{k}': v for k, v in rec.record_frame('ref', ex_ref).items()})
        out.update({f'example/{k}': v for k, v in rec.record_frame('query', ex_query).items()})
        mck = dict(cell_type_col='cell_type', original_idx_col='cell_idx', x_col='X', y_col='Y', max_metacell_size=1, r_max=5, min_angle_deg=5,
                   use_alpha_shape=False, alpha=None, return_object=True)
        ex_mc_a = quiet(ref.metacell_utils.greedy_triangle_collapse, ex_query, **mck)
        ex_mc_r = quiet(ref.metacell_utils.greedy_triangle_collapse, ex_ref, **mck)
        ex_gp = ref.same.init_gurobi_params()
        ex_gp.update(mip_gap=0.025, lazy_allowed_flip_fraction=0.0, time_limit=7200, mip_focus=2, init_method='greedy')
        ex_op = ref.same.init_optim_params()
        ex_op.update({'window_size': 100, 'overlap': 0, 'min_cells_per_window': 30, 'max_matches': 2, 'radius': 5, 'knn': 8,
                      'no_match_penalty': 10000, 'dist_ct_coeff': 1, 'min_angle_deg': 5, 'penalty_coeff': 100, 'delaunay_penalty': 10,
                      'cell_id_col': 'metacell_id', 'ref_metacell_match_multiplier': 1, 'ignore_same_type_triangles': False,
                      'lazy_constraints': True})
        ex_res = quiet(ref.same.sliding_window_matching, ex_mc_r, ex_mc_a, outprefix=os.path.join(work, 'example'), optim_params=ex_op,
                       gurobi_params=ex_gp, ignore_precomputed_triangulation=False)
        out.update({f'example/{k}': v for k, v in rec.record_frame('res', ex_res).items()})
        out.update({f'example/{k}': v for k, v in rec.record_model(fg.Model.last).items()})
        print(f"[shipped example] {len(ex_mc_a.metacell_df)} x {len(ex_mc_r.metacell_df)} metacells, {len(ex_res)} matches, "
              f"{len(fg.Model.last.lazy)} cuts in the last window")

        # sliding windows: tiling, merges of under-populated windows, central trimming, window ids, resume file
        cells = synth.make_cells(1500, 3, seed=51)
        r_big = synth.to_frame(cells)
        m_big = synth.to_frame(synth.make_jittered(cells, seed=52))
        # thin out one corner so that some windows fall under min_cells_per_window and get merged right / down
        m_big = m_big[~((m_big['X'] < 120) & (m_big['Y'] < 170) & (np.arange(len(m_big)) % 4 != 0))].reset_index(drop=True)
        swp = dict(radius=20, knn=4, window_size=150, overlap=40, min_cells_per_window=60)
        res = quiet(ref.same.sliding_window_matching, r_big.copy(), m_big.copy(), commonCT=synth.type_columns(3), outprefix=os.path.join(work, 'sw'),
                    optim_params=dict(swp), gurobi_params=dict(init_method='greedy', lazy_allowed_flip_fraction=0.0))
        out.update({f'sw/{k}': v for k, v in rec.record_frame('res', res).items()})
        out['sw/dirs'] = np.array(sorted(d for d in os.listdir(os.path.join(work, 'sw')) if d.startswith('window_')), dtype=str)
        saved = pd.read_csv(os.path.join(work, 'sw', 'matchedDF.csv'))
        out['sw/csv_rows'] = np.array([len(saved)])
        print(f"[sliding_window] {len(res)} central matches from windows {sorted(res['window_id'].unique().tolist())}")
        # resume: same call again with the CSV present -> nothing left to process, existing rows returned
        res2 = quiet(ref.same.sliding_window_matching, r_big.copy(), m_big.copy(), commonCT=synth.type_columns(3), outprefix=os.path.join(work, 'sw'),
                     optim_params=dict(swp), gurobi_params=dict(init_method='greedy', lazy_allowed_flip_fraction=0.0))
        out['sw/resume_rows'] = np.array([len(res2)])
        # commonCT inferred from cell_type, no outprefix
        res3 = quiet(ref.same.sliding_window_matching, r_big.copy(), m_big.copy(), optim_params=dict(swp, window_size=220, overlap=60),
                     gurobi_params=dict(init_method='greedy', lazy_allowed_flip_fraction=0.0))
        out.update({f'sw_infer/{k}': v for k, v in rec.record_frame('res', res3).items()})
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'run_same_mock.npz'), **out)
    print(f"[run_same_mock] {len(out)} arrays")


def window_tiler_case():
    """sliding_window_matching's tiling / merging / trimming / resume logic (src/same.py:481-590, src/helpers.py:21-70) RUN AS-IS
    with run_same replaced by a recorder that returns every moving cell of the window as a 'match' (so the fixture holds which
    cells each window received, in which order, under which outprefix, and what survived the central trim)."""
    import shutil
    import tempfile
    from run_same_record import tiler_inputs

    cfgs = np.array([
        # n_ref n_mov side seed ws   ov  min hole
        [1200, 1100, 400, 1, 150, 40, 60, 1],
        [900, 900, 300, 2, 100, 0, 30, 0],
        [1500, 1300, 520, 3, 200, 50, 120, 1],
        [400, 380, 250, 4, 120, 30, 40, 1],
        [2000, 1800, 610, 5, 130, 64, 25, 1],
        [300, 300, 99, 6, 1000, 250, 10, 0],      # one window covers everything (the defaults)
        [800, 700, 333, 7, 111, 11, 80, 1],
        [600, 650, 480, 8, 160, 80, 90, 1],
        [50, 40, 200, 9, 100, 20, 30, 0],         # too few cells everywhere: nothing is processed
        [1000, 900, 350.5, 10, 90, 45, 35, 1],
    ], dtype=float)
    out = {'cfgs': cfgs}
    work = tempfile.mkdtemp(prefix='same_golden_')
    calls = []

    def recorder(aligned_df, ref_df, commonCT, optim_params, gurobi_params, outprefix, aligned_delaunay, aligned_delaunay_vertex_col,
                 ignore_precomputed_triangulation):
        calls.append((os.path.basename(outprefix) if outprefix else '', aligned_df['Cell_Num_Old'].to_numpy().copy(),
                      ref_df['Cell_Num_Old'].to_numpy().copy()))
        m = pd.DataFrame({'X': aligned_df['X'].to_numpy(), 'Y': aligned_df['Y'].to_numpy(),
                          'Aligned_Cell_Num_Old': aligned_df['Cell_Num_Old'].to_numpy()})
        return m, {}

    def pack(tag):
        out[f'{tag}/prefix'] = np.array([c[0] for c in calls], dtype=str)
        out[f'{tag}/a_off'] = np.concatenate(([0], np.cumsum([len(c[1]) for c in calls]))).astype(np.int64)
        out[f'{tag}/r_off'] = np.concatenate(([0], np.cumsum([len(c[2]) for c in calls]))).astype(np.int64)
        out[f'{tag}/a_ids'] = np.concatenate([c[1] for c in calls] + [np.zeros(0, np.int64)]).astype(np.int64)
        out[f'{tag}/r_ids'] = np.concatenate([c[2] for c in calls] + [np.zeros(0, np.int64)]).astype(np.int64)
        calls.clear()

    def result(tag, res):
        if len(res) == 0:
            out[f'{tag}/res'] = np.zeros((0, 2), np.int64)
        else:
            out[f'{tag}/res'] = res[['Aligned_Cell_Num_Old', 'window_id']].to_numpy(dtype=np.int64)

    saved = ref.same.run_same
    ref.same.run_same = recorder
    try:
        for q, cfg in enumerate(cfgs):
            r_df, m_df = tiler_inputs(cfg)
            op = dict(window_size=int(cfg[4]), overlap=int(cfg[5]), min_cells_per_window=int(cfg[6]))
            res = quiet(ref.same.sliding_window_matching, r_df.copy(), m_df.copy(), commonCT=['a'], optim_params=dict(op))
            pack(f'c{q}/plain'); result(f'c{q}/plain', res)
            pre = os.path.join(work, f'c{q}')
            res = quiet(ref.same.sliding_window_matching, r_df.copy(), m_df.copy(), commonCT=['a'], outprefix=pre, optim_params=dict(op))
            pack(f'c{q}/out'); result(f'c{q}/out', res)
            csv = os.path.join(pre, 'matchedDF.csv')
            if os.path.exists(csv):          # partial progress: drop every other window id from the file, then resume
                full = pd.read_csv(csv)
                ids = sorted(full['window_id'].unique())
                full[full['window_id'].isin(ids[::2])].to_csv(csv, index=False)
                out[f'c{q}/kept_ids'] = np.array(ids[::2], dtype=np.int64)
                res = quiet(ref.same.sliding_window_matching, r_df.copy(), m_df.copy(), commonCT=['a'], outprefix=pre, optim_params=dict(op))
                pack(f'c{q}/resume'); result(f'c{q}/resume', res)
            print(f"[tiler c{q}] ws={int(cfg[4])} ov={int(cfg[5])} min={int(cfg[6])}: {len(out[f'c{q}/plain/prefix'])} windows run, "
                  f"{len(out[f'c{q}/plain/res'])} central rows; with outprefix {len(out[f'c{q}/out/prefix'])} windows")
    finally:
        ref.same.run_same = saved
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'window_tiler.npz'), **out)


TONGUE_TYPES = ['Endothelial cells', 'Epithelial cells', 'Fibroblasts', 'Lymphoid cells', 'Myeloid cells']


def tongue_case():
    """Real data: examples/tongue (protein section = query, RNA section = template; 64-bit cell ids) driven as its
    run_same.sh drives it (parameter block :34-47, calls :75-121), once with MS=1 and once with MS=3 metacells + unpacking,
    through the recording solver double."""
    import fake_gurobipy as fg
    import run_same_record as rec
    import shutil
    import tempfile

    d = os.path.join(REF_ROOT, 'examples', 'tongue', 'data')
    raw_a = pd.read_csv(os.path.join(d, 'prot_df.csv'), index_col=0)
    raw_r = pd.read_csv(os.path.join(d, 'mer_df.csv'), index_col=0)
    out = {}
    out.update(rec.record_frame('prot', raw_a))
    out.update(rec.record_frame('mer', raw_r))
    work = tempfile.mkdtemp(prefix='same_golden_')
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for ms in (1, 3):
            a_df, r_df = raw_a.copy(), raw_r.copy()
            for df in (a_df, r_df):
                df['X'] = df['transformed_x']
                df['Y'] = df['transformed_y']
                df[TONGUE_TYPES] = df[TONGUE_TYPES] * 100
                df['cell_type'] = df[TONGUE_TYPES].idxmax(axis=1)
            mck = dict(cell_type_col='cell_type', original_idx_col='Cell_Num', x_col='X', y_col='Y', max_metacell_size=ms, r_max=300,
                       min_angle_deg=15, use_alpha_shape=False, return_object=True)
            mc_a = quiet(ref.metacell_utils.greedy_triangle_collapse, a_df, **mck)
            mc_r = quiet(ref.metacell_utils.greedy_triangle_collapse, r_df, **mck)
            gp_ = ref.same.init_gurobi_params()
            gp_.update(mip_gap=0.05, lazy_allowed_flip_fraction=0.05, init_method='greedy')
            op = ref.same.init_optim_params()
            op.update({'window_size': 4000, 'overlap': 300, 'min_cells_per_window': 30, 'max_matches': 1, 'radius': 300, 'knn': 8,
                       'no_match_penalty': 10000, 'penalty_coeff': 100, 'dist_ct_coeff': 1, 'delaunay_penalty': 10,
                       'cell_id_col': 'metacell_id', 'ref_metacell_match_multiplier': ms, 'lazy_constraints': True, 'min_angle_deg': 15})
            res = quiet(ref.same.sliding_window_matching, mc_r, mc_a, commonCT=TONGUE_TYPES, outprefix=os.path.join(work, f'ms{ms}'),
                        optim_params=op, gurobi_params=gp_, ignore_precomputed_triangulation=False)
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_frame('res', res).items()})
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_model(fg.Model.last).items()})
            out[f'ms{ms}/n_metacells'] = np.array([len(mc_a.metacell_df), len(mc_r.metacell_df)])
            indiv = ref.metacell_utils.unpack_metacell_matches(res, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_df, ref_df=r_df,
                                                               strategy='nearest', aligned_original_idx_col='Cell_Num',
                                                               ref_original_idx_col='Cell_Num')
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_frame('unp', indiv[['Aligned_cell_id', 'Ref_cell_id']]).items()})
            print(f"[tongue MS={ms}] {len(mc_a.metacell_df)} x {len(mc_r.metacell_df)} metacells, windows {sorted(res['window_id'].unique().tolist())}, "
                  f"{len(res)} matches -> {len(indiv)} cell matches")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'real_tongue.npz'), **out)


HEART_TYPES = ['Smooth muscle cells', 'Fibroblast', 'Atrial cardiomyocytes', 'Cardiomyocytes', 'Endothelium', 'Epicardium',
               'Schwan progenitors', 'Ventricular cardiomyocytes']


def heart_case():
    """Real data: examples/heart (ISS heart spots; percentages built from small cell counts, so exact zeros and exact cost
    ties abound; spots sit on a 242.5-unit lattice, so Delaunay input is degenerate and equal distances are common) driven as
    its run_same.sh drives it (parameters :39-52, calls :80-126; the `_percentage` suffix of the CSV columns is stripped as the
    example's notebook does), with MS=1 and MS=3, through the recording solver double.  radius and r_max are 500 instead of the
    script's 50: in spot_x/spot_y units 50 is below the lattice spacing, and the reference itself then dies with IndexError at
    src/same.py:1253 on the first window whose nodes are all unconstrained."""
    import fake_gurobipy as fg
    import run_same_record as rec
    import shutil
    import tempfile

    d = os.path.join(REF_ROOT, 'examples', 'heart', 'data')
    raw_a = pd.read_csv(os.path.join(d, 'queryAD_valis.csv'))
    raw_r = pd.read_csv(os.path.join(d, 'refAD_valis.csv'))
    out = {}
    out.update(rec.record_frame('query', raw_a))
    out.update(rec.record_frame('ref', raw_r))
    work = tempfile.mkdtemp(prefix='same_golden_')
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for ms in (1, 3):
            a_df, r_df = raw_a.copy(), raw_r.copy()
            for df in (a_df, r_df):
                df.rename(columns={f'{ct}_percentage': ct for ct in HEART_TYPES}, inplace=True)
                df['X'] = df['spot_x'] + 75
                df['Y'] = df['spot_y'] + 75
                df['cell_type'] = df[HEART_TYPES].idxmax(axis=1)
            mck = dict(cell_type_col='cell_type', original_idx_col='Cell_Num', x_col='X', y_col='Y', max_metacell_size=ms, r_max=500,
                       min_angle_deg=15, use_alpha_shape=False, return_object=True)
            mc_a = quiet(ref.metacell_utils.greedy_triangle_collapse, a_df, **mck)
            mc_r = quiet(ref.metacell_utils.greedy_triangle_collapse, r_df, **mck)
            gp_ = ref.same.init_gurobi_params()
            gp_.update(mip_gap=0.05, lazy_allowed_flip_fraction=0.05, time_limit=7200, init_method='greedy')
            op = ref.same.init_optim_params()
            op.update({'window_size': 4000, 'overlap': 100, 'min_cells_per_window': 30, 'max_matches': 1, 'radius': 500, 'knn': 8,
                       'no_match_penalty': 10000, 'penalty_coeff': 100, 'dist_ct_coeff': 1, 'delaunay_penalty': 10,
                       'cell_id_col': 'metacell_id', 'ref_metacell_match_multiplier': ms, 'ignore_same_type_triangles': True,
                       'lazy_constraints': True, 'min_angle_deg': 15})
            res = quiet(ref.same.sliding_window_matching, mc_r, mc_a, commonCT=HEART_TYPES, outprefix=os.path.join(work, f'ms{ms}'),
                        optim_params=op, gurobi_params=gp_, ignore_precomputed_triangulation=False)
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_frame('res', res).items()})
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_model(fg.Model.last).items()})
            out[f'ms{ms}/n_metacells'] = np.array([len(mc_a.metacell_df), len(mc_r.metacell_df)])
            indiv = ref.metacell_utils.unpack_metacell_matches(res, mc_a.metacell_df, mc_r.metacell_df, aligned_df=a_df, ref_df=r_df,
                                                               strategy='nearest', aligned_original_idx_col='Cell_Num',
                                                               ref_original_idx_col='Cell_Num')
            out.update({f'ms{ms}/{k}': v for k, v in rec.record_frame('unp', indiv[['Aligned_cell_id', 'Ref_cell_id']]).items()})
            print(f"[heart MS={ms}] {len(mc_a.metacell_df)} x {len(mc_r.metacell_df)} metacells, windows {sorted(res['window_id'].unique().tolist())}, "
                  f"{len(res)} matches -> {len(indiv)} cell matches")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'real_heart.npz'), **out)


def run_same_sweep_case(n_cfg=16):
    """run_same through the solver double on seeded random PARAMETER COMBINATIONS (flags x penalties x start methods x lazy/eager):
    per configuration the match table, var_out and the model are recorded exactly as in run_same_mock."""
    import fake_gurobipy as fg
    import run_same_record as rec
    import shutil
    import tempfile

    out = {'n_cfg': np.array([n_cfg])}
    work = tempfile.mkdtemp(prefix='same_golden_')
    cwd = os.getcwd()
    os.chdir(work)
    try:
        for q in range(n_cfg):
            n, T, op, gp_ = rec.random_run_same_config(q)
            cells = synth.make_cells(n, T, seed=700 + q)
            r_df = synth.to_frame(cells)
            a_df = synth.to_frame(synth.make_jittered(cells, seed=800 + q))
            out_df, var_out = quiet(ref.same.run_same, r_df.copy(), a_df.copy(), synth.type_columns(T), outprefix=os.path.join(work, f'c{q}'),
                                    optim_params=ref.same.init_optim_params(**op), gurobi_params=ref.same.init_gurobi_params(**gp_))
            r = rec.record_run(out_df, var_out, fg.Model.last)
            out.update({f'c{q}/{k}': v for k, v in r.items()})
            print(f"[sweep c{q}] n={n} T={T} lazy={op['lazy_constraints']} prio={op['ignore_knn_if_matched']} init={gp_['init_method']} "
                  f"mm={op['max_matches']}: {len(out_df)} matches, {len(fg.Model.last.constrs)} constraints, {len(fg.Model.last.lazy)} cuts")
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)
    np.savez_compressed(os.path.join(OUT, 'run_same_sweep.npz'), **out)


SUBCOMMANDS = ("base", "eval", "metacell", "unpack", "eager", "runsame", "tiler", "sweep", "tongue", "heart")


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 1 and sys.argv[1] == 'all':      # every fixture, one process per family (the solver-facing ones install the double first)
        import subprocess
        for sub in SUBCOMMANDS:
            print(f"== {sub}", flush=True)
            subprocess.run([sys.executable, '-B', os.path.abspath(__file__)] + ([] if sub == 'base' else [sub]), check=True)
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'sweep':
        run_same_sweep_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'heart':
        heart_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'tongue':
        tongue_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'tiler':
        window_tiler_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'eager':
        eager_model_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'runsame':
        run_same_mock_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'unpack':
        unpack_merge_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'mergededup':
        merge_dedup_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'eval':
        eval_case()
        return
    if len(sys.argv) > 1 and sys.argv[1] == 'metacell':
        metacell_case()
        return
    # (1) the shipped synthetic example with the paper's parameters (examples/synthetic/run_same.sh:34-54)
    d = os.path.join(REF_ROOT, 'examples', 'synthetic', 'data')
    ref_df = add_row_ids(pd.read_csv(os.path.join(d, 'ref.csv'), index_col=0))
    qry_df = add_row_ids(pd.read_csv(os.path.join(d, 'query.csv'), index_col=0))
    full_case('synthetic_example', qry_df, ref_df, ['c1', 'c2', 'c3'], radius=5, knn=8, min_angle_deg=5,
              dist_ct_coeff=1, no_match_penalty=10000)
    # (2) BASELINE cfg1 shape: 500x500, T=5, k=8, r=10 on side 100
    r = synth.make_cells(500, 5, seed=0, side=100.0)
    m = synth.make_cells(500, 5, seed=1, side=100.0)
    full_case('cfg1_500', add_row_ids(synth.to_frame(m)), add_row_ids(synth.to_frame(r)), synth.type_columns(5),
              radius=10, knn=8, min_angle_deg=15, dist_ct_coeff=1, no_match_penalty=100)
    # (3) cfg2 shape at reduced size: T=20, k=32, r=25, density 0.01; moving = jittered ref
    r = synth.make_cells(1500, 20, seed=0)
    m = synth.make_jittered(r, seed=1)
    full_case('cfg2_small', add_row_ids(synth.to_frame(m)), add_row_ids(synth.to_frame(r)), synth.type_columns(20),
              radius=25, knn=32, min_angle_deg=15, dist_ct_coeff=1, no_match_penalty=100, cost_sample=3000,
              do_hungarian=True)
    # (4) outputs stored by the reference authors
    stored_run_case('simulated_st')
    stored_run_case('simulated_elastic')
    # (5) adversarial
    adversarial_case()
    # (6) SURVEY 8(f1): eval_utils.check_triangle_violations
    eval_case()
    # (7) SURVEY 8(f2): metacell_utils.greedy_triangle_collapse
    metacell_case()
    # (8) SURVEY 8(f3, f4)
    unpack_merge_case()
    merge_dedup_case()
    sizes = {f: os.path.getsize(os.path.join(OUT, f)) for f in sorted(os.listdir(OUT)) if f.endswith('.npz')}
    print(json.dumps(sizes, indent=1))


if __name__ == '__main__':
    main()
