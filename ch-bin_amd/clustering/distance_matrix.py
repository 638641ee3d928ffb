"""Mirror of ch_bin/core/clustering/distance_matrix.py."""
import logging
import time
from pathlib import Path

import numpy as np
from numpy.lib.format import open_memmap

from .._lib import default_context

logger = logging.getLogger(__name__)


def create_in_mem_distance_matrix(arr: np.ndarray) -> np.ndarray:
    """distance_matrix.py:33-44: dense N x N Euclidean matrix, bit-identical to scipy's cdist.
    (fit_cluster does not need it: the HIP path recomputes distance tiles on the fly.)"""
    start_time = time.time()
    ctx = default_context()
    ctx.set_samples_cached(np.ascontiguousarray(arr, dtype=np.float64))
    result = ctx.pairwise_distance(0, len(arr))
    logger.debug("Distance matrix calculated in %s s.", time.time() - start_time)
    return result


def create_distance_matrix(arr: np.ndarray, operating_dir: Path) -> Path:
    """distance_matrix.py:12-30: same matrix written to <operating_dir>/distance_matrix.npy;
    an existing file is reused without validation, like the reference (:19-22)."""
    n = len(arr)
    filename = Path(operating_dir) / "distance_matrix.npy"
    if filename.exists():
        logger.info("Reusing already existing distance matrix at %s.", filename)
        return filename
    start_time = time.time()
    ctx = default_context()
    ctx.set_samples_cached(np.ascontiguousarray(arr, dtype=np.float64))
    result = open_memmap(filename=filename, mode="w+", shape=(n, n))
    step = max(64, (1 << 27) // max(n, 1))
    for r0 in range(0, n, step):
        r1 = min(n, r0 + step)
        result[r0:r1] = ctx.pairwise_distance(r0, r1)
    result.flush()
    logger.debug("Distance matrix calculated in %s s.", time.time() - start_time)
    return filename


def find_nearest_from_cluster(c: int, curr_bins: np.ndarray, distance_row: np.ndarray, m: int) -> np.ndarray:
    """distance_matrix.py:47-62.  Returns the indices ordered by (distance, index); the reference
    returns the same set in np.argpartition's unspecified order."""
    return default_context().find_nearest_from_row(c, curr_bins, distance_row, m)
