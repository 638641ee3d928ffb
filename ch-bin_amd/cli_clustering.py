"""Mirror of the clustering stage driver ch_bin/cli/clustering.py: features.csv in,
binning-assignment.csv out, same columns, same majority vote, same error for leftovers."""
import logging
from configparser import SectionProxy
from pathlib import Path

import numpy as np
import pandas as pd

from .clustering import fit_cluster
from .clustering.dump_bins import dump_bins

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
    GPU with cdist's rounding.  bins/bin_{i}.fasta (dump_bins.py:8-29) are written when `contig_fasta`
    exists (the reference always has it; tests of the numeric path may pass a placeholder)."""
    operating_dir = Path(operating_dir)
    dist_bin_csv = operating_dir / "binning-assignment.csv"
    operating_dir.mkdir(parents=True, exist_ok=True)
    (operating_dir / "bins").mkdir(parents=True, exist_ok=True)

    # features.csv schema (cli/features.py:96-110): CONTIG_NAME, PARENT_NAME, CLUSTER, then the k-mer
    # and coverage columns.  Everything that is not one of the three bookkeeping columns is a feature.
    logger.info(">> Reading feature CSV...")
    table = pd.read_csv(features_csv)
    seeds = table["CLUSTER"].to_numpy(dtype=np.int64, copy=True)          # -1 = to be binned
    n_bins = int(seeds.max()) + 1                                         # cli/clustering.py:51
    feature_cols = [c for c in table.columns if c not in ("CONTIG_NAME", "PARENT_NAME", "CLUSTER")]
    feats = np.ascontiguousarray(table[feature_cols].to_numpy(dtype=np.float64))

    # cli/clustering.py:55-63 builds an N x N matrix here; the HIP path needs none.
    logger.info(">> Skipping the %s distance matrix (tiles are recomputed on the GPU)...",
                (len(feats), len(feats)))

    logger.info(">> Performing binning using %s solver...", qp_solver)
    labels = fit_cluster(
        samples=feats,
        num_clusters=n_bins,
        distance_matrix=None,
        initial_bins=seeds,
        num_neighbors=num_neighbors,
        max_iterations=max_iterations,
        metric=metric,
        qp_solver=qp_solver,
    )
    if np.any(labels < 0):                                                # cli/clustering.py:79-80
        raise ValueError("There were some un-clustered points left... Aborting.")

    # One row per parent contig: the bin most of its sub-contigs landed in, ties to the lowest
    # bin id, parents in sorted order -- what cli/clustering.py:84-91 gets from
    # groupby("PARENT_NAME") + np.bincount(x).argmax().
    logger.info(">> Assigning bins...")
    parents, which = np.unique(table["PARENT_NAME"].to_numpy().astype(str), return_inverse=True)
    votes = np.zeros((len(parents), max(n_bins, 1)), dtype=np.int64)
    np.add.at(votes, (which, np.asarray(labels, dtype=np.int64)), 1)
    df_bins = pd.DataFrame({"CONTIG_NAME": parents, "BIN": votes.argmax(axis=1)})
    df_bins.to_csv(dist_bin_csv, index=False)
    logger.info("Dumped binning assignment CSV at %s...", dist_bin_csv)

    # cli/clustering.py:94-97
    if contig_fasta is not None and Path(contig_fasta).is_file():
        logger.info(">> Writing binned FASTA files...")
        dump_bins(df_bins, Path(contig_fasta), operating_dir / "bins")
        logger.info("Dumped binned FASTA files to %s...", operating_dir / "bins")
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
