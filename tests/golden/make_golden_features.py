"""Fixture for the feature-assembly mirror: the reference's own parse_coverages
(ch_bin/core/features/coverage.py, importable as it only needs pandas) on a 3-sample table, which
exercises the column-then-row normalisation branch (coverage.py:38-40) that the reference's
one-sample abundance file does not reach.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_features.py
Writes tests/golden/coverages_multi.npz.  preprocess.py / kmer_count.py cannot be imported here
(Biopython is absent; seq2vec is an external tool), so the split rule is tested against its
documented behaviour and the k-mer kernel against the oracle restatement.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True


def main():
    from ch_bin.core.features import coverage as ref_cov

    rng = np.random.default_rng(11)
    n = 40
    raw = rng.lognormal(1.0, 1.2, size=(n, 3))
    raw[3, 1] = 0.0   # a zero entry
    names = np.array([f"NODE_{i}_length_{1000 + 37 * i}" for i in range(n)])
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "abund.tsv")
        with open(path, "w") as fh:
            for i in range(n):
                fh.write(names[i] + "\t" + "\t".join(repr(float(v)) for v in raw[i]) + "\n")
        df = ref_cov.parse_coverages(path)
    np.savez_compressed(os.path.join(HERE, "coverages_multi.npz"), names=names, raw=raw,
                        normalised=df.drop("CONTIG_NAME", axis=1).to_numpy(dtype=np.float64))
    print("written coverages_multi.npz", df.shape)


if __name__ == "__main__":
    main()
