"""Synthetic contig feature matrices.

Mimics what the reference's feature stage writes to features.csv (cli/features.py:96-110):
a k-mer frequency block whose rows sum to 1 (seq2vec fractions, ~1/Dk each) followed by S
coverage columns normalised exactly as ch_bin/core/features/coverage.py:36-41 does (per column,
then per row when S > 1), plus the initial CLUSTER column (seed contigs labelled, rest -1).
"""
import numpy as np


def make_synthetic(N, D=136, B=64, S=1, seed=0, sigma=1.5e-3, n_seed=None, mix=0.0):
    """Returns (X float64[N,D] C-contiguous, initial_bins int64[N], true_bins int64[N]).

    mix in [0,1) pulls every bin centroid towards a common profile (mix=0 is the SURVEY 8(d)
    generator); values near 1 make bins overlap so that sweeps keep moving contigs."""
    rng = np.random.default_rng(seed)
    Dk = D - S
    centroids = rng.dirichlet(5.0 * np.ones(Dk), size=B)
    if mix > 0.0:
        centroids = (1.0 - mix) * centroids + mix * centroids.mean(axis=0, keepdims=True)
    true = rng.integers(0, B, size=N)
    K = np.abs(centroids[true] + rng.normal(0.0, sigma, size=(N, Dk)))
    K /= K.sum(axis=1, keepdims=True)
    bin_mean = rng.lognormal(mean=3.0, sigma=1.0, size=(B, S))
    cov = bin_mean[true] * rng.lognormal(mean=0.0, sigma=0.1, size=(N, S))
    cov = cov / cov.sum(axis=0)  # coverage.py:37
    if S > 1:
        cov = cov / cov.sum(axis=1, keepdims=True)  # coverage.py:39
    X = np.ascontiguousarray(np.hstack([K, cov]), dtype=np.float64)
    if n_seed is None:
        n_seed = max(20, N // (50 * B))
    initial = np.full(N, -1, dtype=np.int64)
    for c in range(B):
        members = np.flatnonzero(true == c)[:n_seed]
        initial[members] = c
    return X, initial, true.astype(np.int64)


def draw_permutations(initial_bins, max_iterations, seed=0):
    """The permutations fit_cluster would draw: ch_bin.py:22 seeds the legacy global RNG with 0 and
    algorithm.py:45 calls np.random.permutation(points_to_assign) once per sweep."""
    pts = np.where(np.asarray(initial_bins) == -1)[0]
    state = np.random.get_state()
    try:
        np.random.seed(seed)
        perms = np.stack([np.random.permutation(pts) for _ in range(max_iterations)]) \
            if len(pts) else np.zeros((max_iterations, 0), dtype=np.int64)
    finally:
        np.random.set_state(state)
    return perms.astype(np.int64)
