"""Per-bin FASTA files: mirror of ch_bin/core/clustering/dump_bins.py on the package's own FASTA
reader (the reference uses Biopython)."""
from pathlib import Path

import pandas as pd

from ..features.fasta import read_fasta, write_record


def dump_bins(df_bins: pd.DataFrame, contig_fasta: Path, operating_dir: Path):
    """dump_bins.py:8-29.  One `bin_<i>.fasta` per bin id occurring in `df_bins` (columns
    CONTIG_NAME, BIN); every record of `contig_fasta` whose id has an assignment is appended to its
    bin's file, header line kept; records without an assignment are dropped."""
    operating_dir = Path(operating_dir)
    assigned = dict(zip(df_bins["CONTIG_NAME"].astype(str), df_bins["BIN"]))
    handles = {b: open(operating_dir / f"bin_{b}.fasta", "w") for b in pd.unique(df_bins["BIN"])}
    try:
        for ident, rest, seq in read_fasta(contig_fasta):
            if ident in assigned:
                write_record(handles[assigned[ident]], ident, seq, description=rest)
    finally:
        for fh in handles.values():
            fh.close()
