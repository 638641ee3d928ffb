"""Mirror of ch_bin/core/clustering/algorithm.py."""
import logging

import numpy as np

from .._lib import default_context
from .solve_qp import check_solver

logger = logging.getLogger(__name__)


def fit_cluster(
    samples: np.ndarray,
    num_clusters: int,
    initial_bins: np.ndarray,
    distance_matrix: np.ndarray = None,
    num_neighbors: int = 15,
    max_iterations: int = 10,
    metric: str = "convex",
    qp_solver: str = "quadprog",
    batch: int = 0,
) -> np.ndarray:
    """algorithm.py:12-76 with identical semantics and return value.

    `distance_matrix` is accepted for signature compatibility and ignored (may be None): the HIP
    path recomputes the needed distances, with cdist's exact rounding, instead of reading an
    N x N matrix.  The per-sweep permutations come from the same legacy global numpy RNG calls as
    algorithm.py:45, so after `np.random.seed(0)` (ch_bin.py:22) the visiting order -- and the RNG
    state left behind -- are the reference's.
    """
    if metric not in ("convex", "affine", "affine-qp"):
        raise NotImplementedError(f"Metric {metric} not implemented")  # hull_distance.py:108
    check_solver(qp_solver)                                              # solve_qp.py:132

    samples = np.ascontiguousarray(samples, dtype=np.float64)
    initial = np.ascontiguousarray(initial_bins, dtype=np.int64)
    points_to_assign = np.where(initial == -1)[0]                       # algorithm.py:38
    logger.debug("Assigning %s points.", len(points_to_assign))

    # algorithm.py:45 draws one permutation per sweep that actually runs.  Draw them all up front
    # for the kernel, then rewind and replay exactly as many draws as sweeps ran.
    state = np.random.get_state()
    perms = np.stack([np.random.permutation(points_to_assign) for _ in range(max_iterations)]) \
        if max_iterations > 0 else np.zeros((0, len(points_to_assign)), dtype=np.int64)

    ctx = default_context()
    ctx.set_samples_cached(samples)
    with ctx.using_metric(metric):
        labels, iters, changed = ctx.fit_cluster(int(num_clusters), initial, perms.astype(np.int64),
                                                 int(num_neighbors), int(max_iterations), batch=batch)

    np.random.set_state(state)
    for _ in range(iters):
        np.random.permutation(points_to_assign)

    n = len(samples)
    for i_iter in range(iters):                                          # algorithm.py:63-69
        if changed[i_iter] == 0:
            logger.info("Iteration %s: No changes with previous iteration... Stopping...", i_iter + 1)
            break
        logger.info("Iteration %s: Points changed clusters. avg=%s, count=%s", i_iter + 1,
                    changed[i_iter] / n, int(changed[i_iter]))
    else:
        logger.info("Exit due to max iteration limit.")                  # algorithm.py:74-75
    return labels
