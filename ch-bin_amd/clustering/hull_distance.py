"""Mirror of ch_bin/core/clustering/hull_distance.py (convex metric)."""
import numpy as np

from .._lib import default_context
from .solve_qp import check_solver


def convex_hull_distance(query: np.ndarray, points: np.ndarray, solver: str = "quadprog") -> float:
    """hull_distance.py:7-35: distance from `query` to the convex hull of the rows of `points`."""
    check_solver(solver)
    points = np.asarray(points, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    return float(default_context().hull_distance_points(query, points.reshape(-1, query.shape[0])))


def calculate_distance(x: np.ndarray, mat_p: np.ndarray, qp_solver: str, metric: str) -> float:
    """hull_distance.py:90-108.  Only metric='convex' (AlgoDistanceMetric default, default.ini:18)
    is on the accelerated path; the reference's 'affine'/'affine-qp' variants are not built yet."""
    if metric == "convex":
        return convex_hull_distance(x, mat_p, solver=qp_solver)
    raise NotImplementedError(f"Metric {metric} not implemented")  # hull_distance.py:108
