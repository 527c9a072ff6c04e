"""Parameter dictionaries, same keys and defaults as the reference
(init_gurobi_params: src/same.py:40-130, init_optim_params: src/same.py:133-242)."""
from typing import Any, Dict


def init_gurobi_params(**overrides) -> Dict[str, Any]:
    params = {
        "time_limit": 7200,
        "mip_gap": 0.05,
        "mip_focus": 2,
        "cuts": 2,
        "heuristics": 0.1,
        "init_method": None,
        "init_big_m": 1e9,
        "init_hungarian_max_n": 5000,
        "lazy_max_cuts": None,
        "lazy_allowed_flip_fraction": 0.05,
        "lazy_max_cuts_per_incumbent": 1000,
    }
    params.update(overrides)
    return params


def init_optim_params(**overrides) -> Dict[str, Any]:
    params = {
        "window_size": 1000,
        "overlap": 250,
        "min_cells_per_window": 10,
        "max_matches": 1,
        "ref_metacell_match_multiplier": None,
        "radius": 250,
        "penalty_coeff": 100,
        "no_match_penalty": 100,
        "delaunay_penalty": 5,
        "dist_ct_coeff": 1,
        "knn": 8,
        "cell_id_col": "Cell_Num_Old",
        "hard_spatial_constraints": False,
        "ignore_same_type_triangles": True,
        "ignore_knn_if_matched": False,
        "lazy_constraints": True,
        "min_angle_deg": 15,
    }
    params.update(overrides)
    return params
