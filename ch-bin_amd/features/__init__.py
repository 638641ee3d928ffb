"""Feature assembly next to the hot path: mirror of ch_bin.core.features for
the parts that need no external bioinformatics tool -- canonical k-mer frequencies (HIP kernel in
place of the seq2vec run), coverage normalisation, contig filtering / splitting."""
from .coverage import parse_coverages  # noqa: F401
from .kmer_count import count_kmers, kmer_frequencies  # noqa: F401
from .preprocess import filter_short_contigs, get_contig_lengths, split_contigs  # noqa: F401
