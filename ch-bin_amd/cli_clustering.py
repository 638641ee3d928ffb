"""Mirror of the clustering stage driver ch_bin/cli/clustering.py: features.csv in,
binning-assignment.csv out, same columns, same majority vote, same error for leftovers."""
import logging
from configparser import SectionProxy
from pathlib import Path

import numpy as np
import pandas as pd

from .clustering import fit_cluster

logger = logging.getLogger(__name__)


def perform_clustering(
    contig_fasta: Path,
    features_csv: Path,
    operating_dir: Path,
    num_neighbors: int = 15,
    max_iterations: int = 10,
    metric: str = "convex",
    qp_solver: str = "quadprog",
    in_mem_dist_matrix: bool = True,
) -> Path:
    """cli/clustering.py:19-99.  `in_mem_dist_matrix` (InMemDistMatrix) is accepted and both values
    are legal, but no N x N matrix is built: fit_cluster recomputes the distances it needs on the
    GPU with cdist's rounding.  Writing bins/bin_{i}.fasta (dump_bins.py:8-29, needs Biopython) is
    outside the accelerated path and skipped."""
    operating_dir = Path(operating_dir)
    dist_bin_csv = operating_dir / "binning-assignment.csv"
    operating_dir.mkdir(parents=True, exist_ok=True)
    (operating_dir / "bins").mkdir(parents=True, exist_ok=True)

    # 01. Read feature CSV                                               (cli/clustering.py:47-53)
    logger.info(">> Reading feature CSV...")
    df_features = pd.read_csv(features_csv)
    num_clusters = df_features.CLUSTER.max() + 1
    initial_bins: np.ndarray = df_features.CLUSTER.values.copy()
    samples: np.ndarray = df_features.drop(["CONTIG_NAME", "PARENT_NAME", "CLUSTER"], axis=1).values
    num_samples = len(samples)

    # 02. (no distance matrix)                                           (cli/clustering.py:55-63)
    logger.info(">> Skipping the %s distance matrix (tiles are recomputed on the GPU)...",
                (num_samples, num_samples))

    # 03. Perform binning                                                (cli/clustering.py:65-76)
    logger.info(">> Performing binning using %s solver...", qp_solver)
    convex_labels = fit_cluster(
        samples=samples,
        num_clusters=int(num_clusters),
        distance_matrix=None,
        initial_bins=initial_bins,
        num_neighbors=num_neighbors,
        max_iterations=max_iterations,
        metric=metric,
        qp_solver=qp_solver,
    )
    if np.any(convex_labels < 0):                                      # cli/clustering.py:79-80
        raise ValueError("There were some un-clustered points left... Aborting.")

    # 04. Majority vote per parent contig                               (cli/clustering.py:82-92)
    logger.info(">> Assigning bins...")
    df_samples: pd.DataFrame = df_features.drop("CLUSTER", axis=1)
    df_bin_column: pd.DataFrame = pd.DataFrame({"BIN": convex_labels})
    df_combined: pd.DataFrame = pd.concat([df_samples, df_bin_column], axis=1)
    parent_groups = df_combined[["PARENT_NAME", "BIN"]].groupby("PARENT_NAME")
    df_dist_bin: pd.DataFrame = parent_groups.BIN.apply(lambda x: np.bincount(x).argmax()).reset_index()
    df_dist_bin.rename(columns={"PARENT_NAME": "CONTIG_NAME"}, inplace=True)
    df_dist_bin.to_csv(dist_bin_csv, index=False)
    logger.info("Dumped binning assignment CSV at %s...", dist_bin_csv)
    return dist_bin_csv


def run_perform_clustering(contig_fasta: Path, features_csv: Path, operating_dir: Path,
                           parameters: SectionProxy) -> Path:
    """cli/clustering.py:102-127: same INI keys (config/default.ini:16-20)."""
    return perform_clustering(
        contig_fasta=contig_fasta,
        features_csv=features_csv,
        operating_dir=operating_dir,
        num_neighbors=int(parameters["AlgoNumNeighbors"]),
        max_iterations=int(parameters["AlgoMaxIterations"]),
        metric=parameters["AlgoDistanceMetric"],
        qp_solver=parameters["AlgoQpSolver"],
        in_mem_dist_matrix=parameters.getboolean("InMemDistMatrix"),
    )
