"""chbin_amd -- MI355X-native convex-hull binning hot path of CH-Bin.

Mirrors the reference's `ch_bin.core.clustering` call surface (fit_cluster, calculate_distance,
find_nearest_from_cluster, create_in_mem_distance_matrix, ...) on top of a C-ABI HIP library
(include/chbin_hip.h, csrc/).  There is no CPU fallback: every compute entry point raises if the
HIP library or a GPU is missing.
"""
from . import synth  # noqa: F401

__all__ = ["synth"]
