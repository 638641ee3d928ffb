"""Host-side mirror of the reference package `ch_bin.core.clustering` (same function names,
argument meaning and error behaviour) on top of libchbin_hip.so."""
from .algorithm import fit_cluster  # noqa: F401
from .distance_matrix import (  # noqa: F401
    create_distance_matrix,
    create_in_mem_distance_matrix,
    find_nearest_from_cluster,
)
from .dump_bins import dump_bins  # noqa: F401
from .hull_distance import (  # noqa: F401
    affine_hull_distance,
    affine_hull_distance_qp,
    calculate_distance,
    convex_hull_distance,
)
from .solve_qp import SOLVERS, solve_qp  # noqa: F401
