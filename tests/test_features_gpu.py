"""Feature assembly on the GPU: the HIP k-mer kernel through the C ABI against
the oracle restatement -- counts and frequencies bit-exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import chbin_amd  # noqa: F401
    from chbin_amd import _lib
    return _lib.default_context()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _random_contigs(rng, lengths, dirty=True):
    out = []
    for n in lengths:
        s = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=n).copy()
        if dirty and n > 20:
            s[rng.integers(0, n, size=max(1, n // 300))] = ord("N")        # ambiguity codes
            lo = rng.integers(0, n - 10)
            s[lo:lo + 10] |= 0x20                                            # a soft-masked (lower case) run
        out.append(s.tobytes())
    return out


@pytest.mark.parametrize("k", [1, 2, 3, 4, 5, 6, 7])
def test_kmer_kernel_matches_oracle(ctx, O, k):
    rng = np.random.default_rng(100 + k)
    # empty, shorter than k, exactly k, around the 4096-window chunk size, several chunks
    lengths = [0, 1, k - 1, k, k + 1, 50, 1000, 4095, 4096, 4097, 4096 + k - 1, 9000, 40000, 123457]
    seqs = _random_contigs(rng, [max(0, n) for n in lengths])
    seqs += [b"N" * 500, b"ACGT" * 2000, b"a" * 4100 + b"T" * 4100]
    freq, counts = ctx.kmer_frequencies(seqs, k, return_counts=True)
    want_f, want_c = O.kmer_frequencies(seqs, k)
    assert freq.shape == (len(seqs), O.kmer_dim(k))
    assert np.array_equal(counts.astype(np.int64), want_c)
    assert np.array_equal(freq, want_f)
    rows = want_c.sum(axis=1)
    assert np.allclose(freq.sum(axis=1)[rows > 0], 1.0) and not freq[rows == 0].any()


def test_kmer_many_contigs(ctx, O):
    """A metagenome-shaped input: thousands of contigs of 1-30 kb (config/default.ini:12,15)."""
    rng = np.random.default_rng(9)
    lengths = rng.integers(1000, 30000, size=3000)
    seqs = _random_contigs(rng, lengths)
    freq = ctx.kmer_frequencies(seqs, 4)
    want, _ = O.kmer_frequencies(seqs[:400], 4)
    assert freq.shape == (3000, 136) and np.array_equal(freq[:400], want)


def test_kmer_errors(ctx):
    from chbin_amd._lib import ChbError
    with pytest.raises(ChbError):
        ctx.kmer_frequencies([b"ACGT"], 8)
    assert ctx.kmer_frequencies([], 4).shape == (0, 136)


def test_count_kmers_mirror(ctx, O, tmp_path):
    """kmer_count.py:110-125 through the mirror: file order kept, CONTIG_NAME column, cache reuse."""
    from chbin_amd.features import fasta, kmer_count
    rng = np.random.default_rng(3)
    seqs = _random_contigs(rng, [1500, 2200, 12000], dirty=False)
    names = ["k141_7_S0", "k141_9_S0", "k141_9_S1"]
    fa = tmp_path / "split-contigs.fasta"
    with open(fa, "w") as fh:
        for n, s in zip(names, seqs):
            fasta.write_record(fh, n, s.decode())
    work = tmp_path / "kmers"
    df = kmer_count.count_kmers(fa, work, k=4, tool="seq2vec")
    assert list(df["CONTIG_NAME"]) == names and df.shape == (3, 137)
    want, _ = O.kmer_frequencies(seqs, 4)
    assert np.array_equal(df.drop("CONTIG_NAME", axis=1).to_numpy(), want)
    assert os.path.exists(work / "normalized_kmer_4.csv")                     # kmer_count.py:76
    again = kmer_count.count_kmers(tmp_path / "missing.fasta", work, k=4, tool="seq2vec")   # cache hit (:79-81)
    assert np.allclose(again.drop("CONTIG_NAME", axis=1).to_numpy(), want, rtol=0, atol=1e-15)
    df2 = kmer_count.count_kmers(fa, tmp_path / "kc", k=4, tool="kmer_counter")
    assert list(df2.columns[:3]) == ["AAAA", "AAAC", "AAAG"] and list(df2["CONTIG_NAME"]) == names
