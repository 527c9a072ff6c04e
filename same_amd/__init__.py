"""same_amd -- MI355X-native pre-MIP path of SAME (rohitsinghlab/SAME) behind the reference's
own Python signatures.

Public names follow the reference package (src/__init__.py:54-65) for everything on the path:
init_gurobi_params, init_optim_params, run_same, sliding_window_matching; plus the path's
building blocks under their reference names (find_knn_within_radius,
filter_triangles_by_radius, verify_spatial_preservation, compute_mip_start_pairs, ...).
Compute runs in libsame_hip.so (hand-written HIP for gfx950) through a ctypes C ABI
(include/same_hip.h); there is no CPU fallback -- a missing library or GPU raises.
"""
from .params import init_gurobi_params, init_optim_params  # noqa: F401
from .knn import find_knn_within_radius, find_knn_with_cell_type_priority  # noqa: F401
from .cost import pair_costs, dense_cost_matrix  # noqa: F401
from .triangles import (filter_triangles_by_radius, precompute_triangle_info, precompute_coordinate_maps,  # noqa: F401
                        triangle_weights_and_signs, build_simplex_map)
from .sweeps import (LazyOrientationSweep, verify_spatial_preservation, print_violation_report,  # noqa: F401
                     triangle_area_flips)
from .init_helpers import compute_mip_start_pairs, apply_mip_start  # noqa: F401
from .api import prepare_same_inputs, run_same  # noqa: F401
from .window_api import iter_prepared_windows, resident_frames, sliding_window_matching, subset_data  # noqa: F401
from .incumbent import sliding_window_incumbent  # noqa: F401
from .windows import window_plan  # noqa: F401
from .metacell_utils import MetaCell, greedy_triangle_collapse, unpack_metacell_matches  # noqa: F401
from .merge import merge_window_matches_unique_ref, load_matching_results  # noqa: F401
from .eval_utils import check_triangle_violations  # noqa: F401

__version__ = "0.1.0"
__all__ = [
    "init_gurobi_params", "init_optim_params", "sliding_window_matching", "sliding_window_incumbent", "run_same", "prepare_same_inputs",
    "find_knn_within_radius", "find_knn_with_cell_type_priority", "pair_costs", "dense_cost_matrix",
    "filter_triangles_by_radius", "precompute_triangle_info", "triangle_weights_and_signs",
    "LazyOrientationSweep", "verify_spatial_preservation", "print_violation_report", "triangle_area_flips",
    "compute_mip_start_pairs", "apply_mip_start", "window_plan", "MetaCell", "greedy_triangle_collapse",
    "check_triangle_violations", "unpack_metacell_matches", "merge_window_matches_unique_ref", "load_matching_results",
]
