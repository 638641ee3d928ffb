"""Contig length filter and seed-contig splitting: mirror of ch_bin/core/features/preprocess.py
(same names, arguments and return values), on an own FASTA reader."""
from pathlib import Path
from typing import Dict, List

from .fasta import read_fasta, write_record


def _piece_bounds(length: int, split_len: int):
    """Start/end of the pieces of a sequence (preprocess.py:17-35): pieces of exactly split_len,
    except that the last one absorbs a remainder shorter than split_len -- so it is between
    split_len and 2 * split_len long whenever the sequence is at least split_len long."""
    start = 0
    while start < length:
        if start + 2 * split_len > length:
            yield start, length
            return
        yield start, start + split_len
        start += split_len


def split_contigs(input_fasta: Path, output_fasta: Path, split_contig_ids: List[str],
                  split_len: int = 10000) -> Dict[str, str]:
    """preprocess.py:38-67.  Every record is written as `<id>_S<i>` (description dropped); records
    whose id is listed are cut into pieces first.  Returns {sub contig id: parent id}."""
    wanted = set(split_contig_ids)
    parents: Dict[str, str] = {}
    with open(output_fasta, "w") as out:
        for ident, _rest, seq in read_fasta(input_fasta):
            bounds = list(_piece_bounds(len(seq), split_len)) if ident in wanted else [(0, len(seq))]
            for i, (b, e) in enumerate(bounds):
                sub_id = f"{ident}_S{i}"
                parents[sub_id] = ident
                write_record(out, sub_id, seq[b:e])
    return parents


def filter_short_contigs(input_fasta: Path, output_fasta: Path, threshold: int = 1000) -> List[str]:
    """preprocess.py:70-88: copy records of at least `threshold` bases, return the ids of the rest."""
    removed: List[str] = []
    with open(output_fasta, "w") as out:
        for ident, rest, seq in read_fasta(input_fasta):
            if len(seq) >= threshold:
                write_record(out, ident, seq, description=rest)
            else:
                removed.append(ident)
    return removed


def get_contig_lengths(input_fasta: Path) -> Dict[str, int]:
    """preprocess.py:91-101."""
    return {ident: len(seq) for ident, _rest, seq in read_fasta(input_fasta)}
