"""Normalised canonical k-mer frequencies: mirror of ch_bin/core/features/kmer_count.py with the
external tool run (seq2vec / kmer-counter) replaced by the HIP kernel behind chb_kmer_frequencies."""
import itertools
import logging
from pathlib import Path

import numpy as np
import pandas as pd

from .. import _lib
from .fasta import read_fasta

logger = logging.getLogger(__name__)


def kmer_frequencies(sequences, k: int = 4, device=None, return_counts: bool = False):
    """[n, dim] float64 rows of count / total over the canonical k-mers of each sequence (bytes or
    str).  Columns in the order of canonical_kmers(k)."""
    return _lib.default_context(device).kmer_frequencies(sequences, k, return_counts=return_counts)


def canonical_kmers(k: int):
    """Column labels: the canonical k-mers in column order (smaller 2-bit code of the k-mer and its
    reverse complement, A<C<G<T, ascending)."""
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    out = []
    for tup in itertools.product("ACGT", repeat=k):
        s = "".join(tup)
        rc = "".join(comp[ch] for ch in reversed(s))
        if s <= rc:
            out.append(s)
    return out


def count_kmers(contig_fasta: Path, operating_dir: Path, k: int = 4, tool: str = "kmer_counter") -> pd.DataFrame:
    """kmer_count.py:110-125.  Returns one row per record of `contig_fasta` (in file order) with a
    CONTIG_NAME column and the normalised k-mer columns.  `tool` keeps the reference's values:
    "seq2vec" -> integer column labels 0..dim-1 and the reuse-if-exists cache
    `normalized_kmer_<k>.csv` (kmer_count.py:76-81,99-105); "kmer_counter" -> k-mer strings as column
    labels and `normalized_kmer.csv` (kmer_count.py:29-62); anything else -> NotImplementedError."""
    operating_dir = Path(operating_dir)
    if tool == "seq2vec":
        cache = operating_dir / f"normalized_kmer_{k}.csv"
        if cache.exists():
            logger.info("Found previous run, skipping k-mer counting for k=%s.", k)
            return pd.read_csv(cache)
        labels = None
    elif tool == "kmer_counter":
        logger.warning("kmer-counter mode is deprecated. Use seq2vec instead.")
        cache = operating_dir / "normalized_kmer.csv"
        labels = canonical_kmers(k)
    else:
        raise NotImplementedError(f"Tool {tool} is not implemented")
    names, seqs = [], []
    for ident, _rest, seq in read_fasta(contig_fasta):
        names.append(ident)
        seqs.append(seq)
    logger.debug("Found %s contig names in contig file.", len(names))
    freq = kmer_frequencies(seqs, k) if names else np.zeros((0, len(canonical_kmers(k))))
    df = pd.DataFrame(freq, columns=labels if labels is not None else list(range(freq.shape[1])))
    df["CONTIG_NAME"] = names
    operating_dir.mkdir(parents=True, exist_ok=True)
    df.to_csv(cache, index=False)
    return df
