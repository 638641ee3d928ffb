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


def _distance(query, points, metric):
    points = np.asarray(points, dtype=np.float64)
    query = np.asarray(query, dtype=np.float64)
    with default_context().using_metric(metric) as ctx:
        return float(ctx.hull_distance_points(query, points.reshape(-1, query.shape[0])))


def affine_hull_distance(query: np.ndarray, points: np.ndarray) -> float:
    """hull_distance.py:69-87: distance from `query` to the affine hull of the rows of `points`."""
    return _distance(query, points, "affine")


def affine_hull_distance_qp(query: np.ndarray, points: np.ndarray, solver: str = "quadprog") -> float:
    """hull_distance.py:38-66: the same distance posed as an equality-only QP."""
    check_solver(solver)
    return _distance(query, points, "affine-qp")


def calculate_distance(x: np.ndarray, mat_p: np.ndarray, qp_solver: str, metric: str) -> float:
    """hull_distance.py:90-108: same dispatch, same NotImplementedError for an unknown metric."""
    if metric == "convex":
        return convex_hull_distance(x, mat_p, solver=qp_solver)
    if metric == "affine":
        return affine_hull_distance(x, mat_p)
    if metric == "affine-qp":
        return affine_hull_distance_qp(x, mat_p, solver=qp_solver)
    raise NotImplementedError(f"Metric {metric} not implemented")  # hull_distance.py:108
