"""Fixture for the feature-assembly mirror: the reference's own parse_coverages
(ch_bin/core/features/coverage.py, importable as it only needs pandas) on a 3-sample table, which
exercises the column-then-row normalisation branch (coverage.py:38-40) that the reference's
one-sample abundance file does not reach.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden_features.py
Writes tests/golden/coverages_multi.npz and tests/golden/preprocess.npz.

preprocess.npz: the reference's own preprocess.py (17-101) imported unchanged.  Biopython is absent from
this image, so this script injects an in-process stand-in `Bio` package *into this process only*: `SeqIO.parse` hands the
reference records (id, sequence) read by a six-line FASTA loop below, `SeqIO.write` RECORDS what the reference asks to be
written, `SeqRecord` is a plain holder.  What is pinned is therefore the reference's logic between reader and writer --
`_generate_split_string`'s piece boundaries (a pure function, no stand-in involved), which records are split, the
`_S{i}` names, the dropped description, the parent map, the `>= threshold` keep rule, the length map -- and NOT
Biopython's own header parsing or line wrapping.  kmer_count.py runs an external tool (seq2vec) whose source is not in
the reference tree: the k-mer kernel is tested against the oracle restatement.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True


RECORDED = []   # (id, description, sequence) of every SeqIO.write call


def _install_bio_standin():
    import types

    class SeqRecord:
        def __init__(self, seq, id="", description=""):
            self.seq, self.id, self.description = seq, id, description

    def parse(handle, fmt):
        assert fmt == "fasta"
        ident, desc, chunks = None, "", []
        for line in handle:
            line = line.rstrip("\r\n")
            if line.startswith(">"):
                if ident is not None:
                    yield SeqRecord("".join(chunks), ident, desc)
                head = line[1:].strip()
                ident, desc, chunks = (head.split(None, 1) + [""])[0], head, []
            elif ident is not None:
                chunks.append(line.strip())
        if ident is not None:
            yield SeqRecord("".join(chunks), ident, desc)

    def write(record, handle, fmt):
        assert fmt == "fasta"
        RECORDED.append((str(record.id), str(record.description), str(record.seq)))
        handle.write(">" + str(record.id) + "\n" + str(record.seq) + "\n")
        return 1

    bio = types.ModuleType("Bio")
    seqio = types.ModuleType("Bio.SeqIO")
    seqio.parse, seqio.write = parse, write
    seqrecord = types.ModuleType("Bio.SeqRecord")
    seqrecord.SeqRecord = SeqRecord
    bio.SeqIO, bio.SeqRecord = seqio, seqrecord
    sys.modules["Bio"], sys.modules["Bio.SeqIO"], sys.modules["Bio.SeqRecord"] = bio, seqio, seqrecord


def preprocess_fixture():
    _install_bio_standin()
    from ch_bin.core.features import preprocess as ref_pre

    # (a) the split rule itself: piece lengths for a table of (length, split_len)
    table = []
    for split_len in (1, 7, 1000, 10000):
        for length in sorted({0, 1, split_len - 1, split_len, split_len + 1, 2 * split_len - 1, 2 * split_len,
                              2 * split_len + 1, 3 * split_len - 1, 3 * split_len, 5 * split_len + split_len // 2}):
            if length < 0:
                continue
            pieces = list(ref_pre._generate_split_string("x" * length, split_len))
            table.append((length, split_len, [len(p) for p in pieces]))
    maxp = max(len(t[2]) for t in table)
    piece_len = np.full((len(table), maxp), -1, dtype=np.int64)
    for i, t in enumerate(table):
        piece_len[i, : len(t[2])] = t[2]

    # (b) the three file-level functions on a small FASTA (descriptions, a wrapped sequence, a contig listed for
    # splitting that is shorter than the split length, one that is not listed)
    rng = np.random.default_rng(5)
    lens = {"seedA": 2600, "seedB": 999, "plain": 2100, "tiny": 40, "edge": 1000}
    recs = [(k, "len=%d cov=%.1f" % (v, rng.random() * 9), "".join(rng.choice(list("ACGT"), v))) for k, v in lens.items()]
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "in.fa")
        with open(src, "w") as fh:
            for ident, desc, seq in recs:
                fh.write(">" + ident + " " + desc + "\n")
                for i in range(0, len(seq), 70):
                    fh.write(seq[i:i + 70] + "\n")
        lengths = ref_pre.get_contig_lengths(src)
        RECORDED.clear()
        removed = ref_pre.filter_short_contigs(src, os.path.join(td, "f.fa"), threshold=1000)
        kept = [(i, s) for i, _d, s in RECORDED]
        RECORDED.clear()
        parents = ref_pre.split_contigs(src, os.path.join(td, "s.fa"), ["seedA", "seedB", "edge"], split_len=1000)
        split_written = list(RECORDED)
    np.savez_compressed(
        os.path.join(HERE, "preprocess.npz"),
        table_length=np.array([t[0] for t in table]), table_split_len=np.array([t[1] for t in table]), table_pieces=piece_len,
        in_ids=np.array([r[0] for r in recs]), in_desc=np.array([r[1] for r in recs]), in_seq=np.array([r[2] for r in recs]),
        length_ids=np.array(list(lengths.keys())), length_vals=np.array(list(lengths.values()), dtype=np.int64),
        removed=np.array(removed), kept_ids=np.array([k[0] for k in kept]), kept_seq=np.array([k[1] for k in kept]),
        split_ids=np.array([w[0] for w in split_written]), split_desc=np.array([w[1] for w in split_written]),
        split_seq=np.array([w[2] for w in split_written]),
        parent_keys=np.array(list(parents.keys())), parent_vals=np.array(list(parents.values())))
    print("written preprocess.npz:", len(table), "split-rule rows,", len(split_written), "split records")


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
    preprocess_fixture()


if __name__ == "__main__":
    main()
