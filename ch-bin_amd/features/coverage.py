"""Coverage / abundance table: mirror of ch_bin/core/features/coverage.py."""
from pathlib import Path

import pandas as pd


def parse_coverages(coverage_file: Path, delimiter: str = "\t") -> pd.DataFrame:
    """coverage.py:14-43.  First column = contig name (renamed CONTIG_NAME), the others one coverage
    value per sample.  Every sample column is divided by its sum; with more than one sample every
    row is then divided by its sum (coverage.py:35-40)."""
    table = pd.read_csv(coverage_file, sep=delimiter, header=None).rename(columns={0: "CONTIG_NAME"})
    sample_cols = [c for c in table.columns if c != "CONTIG_NAME"]
    values = table[sample_cols]
    values = values.div(values.sum(axis=0), axis=1)
    if len(sample_cols) > 1:
        values = values.div(values.sum(axis=1), axis=0)
    table[sample_cols] = values
    return table
