"""BASELINE.json configs[4] through the clustering-stage driver: features.csv -> perform_clustering ->
binning-assignment.csv + bins/bin_<i>.fasta (cli/clustering.py:47-97, dump_bins.py:8-29), with wall-clock per stage.

    python tools/cli_config4.py [--contigs 1000000] [--dim 146] [--bins 200] [--neighbors 5] [--work /tmp/chb_cfg4]
                                [--out profiles/r03_cli_config4.json]

The features are the synthetic generator's (the reference's own pipeline needs FragGeneScan / HMMER / seq2vec and the
missing FASTA blob); contigs are sub-contigs `<parent>_S<k>` of synthetic parents as preprocess.py:72-101 names them, so
that the per-parent majority vote has something to vote on.  Checks: every parent's BIN equals an independent
recomputation of the vote; the converged labels are a fixed point of the reference sweep on an oracle sample.
"""
import argparse
import json
import logging
import os
import sys
import time

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StageClock(logging.Handler):
    """perform_clustering announces its stages through the reference's log lines: time between them."""

    def __init__(self):
        super().__init__()
        self.marks = []

    def emit(self, record):
        self.marks.append((time.perf_counter(), record.getMessage()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--contigs", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=146)
    ap.add_argument("--bins", type=int, default=200)
    ap.add_argument("--neighbors", type=int, default=5)          # config/default.ini:16
    ap.add_argument("--sub", type=int, default=4, help="sub-contigs per parent contig")
    ap.add_argument("--work", default="/tmp/chb_cfg4")
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r03_cli_config4.json"))
    ap.add_argument("--oracle-sample", type=int, default=256)
    args = ap.parse_args()

    import chbin_amd
    from chbin_amd import cli_clustering, synth
    from oracle import oracle as O

    N, D, B, m = args.contigs, args.dim, args.bins, args.neighbors
    S = 1 if D <= 136 else (5 if D == 140 else 10)
    os.makedirs(args.work, exist_ok=True)
    t0 = time.perf_counter()
    X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
    rng = np.random.default_rng(1)
    order = rng.permutation(N)                       # sub-contigs of a parent are not neighbours in the table
    parent_of = np.empty(N, dtype=np.int64)
    parent_of[order] = np.arange(N) // args.sub
    sub_of = np.empty(N, dtype=np.int64)
    sub_of[order] = np.arange(N) % args.sub
    parents = np.char.add("contig", parent_of.astype(str))
    names = np.char.add(np.char.add(parents, "_S"), sub_of.astype(str))
    cols = {"CONTIG_NAME": names, "PARENT_NAME": parents, "CLUSTER": initial}
    for k in range(D - S):
        cols[f"KMER_{k}"] = X[:, k]
    for k in range(S):
        cols[f"COV_{k}"] = X[:, D - S + k]
    features_csv = os.path.join(args.work, "features.csv")
    pd.DataFrame(cols).to_csv(features_csv, index=False)
    n_par = int(parent_of.max()) + 1
    fasta = os.path.join(args.work, "contigs.fasta")
    with open(fasta, "w") as fh:
        bases = np.array(list("ACGT"))
        for pid in range(n_par):
            fh.write(f">contig{pid} synthetic\n")
            fh.write("".join(bases[rng.integers(0, 4, 90)]) + "\n")
    t_gen = time.perf_counter() - t0

    clock = StageClock()
    log = logging.getLogger("chbin_amd")
    log.setLevel(logging.INFO)
    log.addHandler(clock)
    for name in ("chbin_amd.cli_clustering", "chbin_amd.clustering.algorithm"):
        logging.getLogger(name).setLevel(logging.INFO)
    captured = []
    orig_fit = cli_clustering.fit_cluster

    def spy(**kw):
        ta = time.perf_counter()
        lab = orig_fit(**kw)
        captured.append((lab, time.perf_counter() - ta))
        return lab

    cli_clustering.fit_cluster = spy
    np.random.seed(0)                                # ch_bin.py:22
    t1 = time.perf_counter()
    out_csv = cli_clustering.perform_clustering(fasta, features_csv, os.path.join(args.work, "out"),
                                                num_neighbors=m, max_iterations=10)
    t_total = time.perf_counter() - t1
    cli_clustering.fit_cluster = orig_fit
    labels, t_fit = captured[0]

    def at(prefix):
        for t, msg in clock.marks:
            if msg.startswith(prefix):
                return t
        return None

    t_read0, t_skip = at(">> Reading feature CSV"), at(">> Skipping")
    t_assign, t_dumped = at(">> Assigning bins"), at("Dumped binning assignment CSV")
    t_fa0, t_fa1 = at(">> Writing binned FASTA"), at("Dumped binned FASTA")
    stages = {
        "read_features_csv_s": t_skip - t_read0,
        "fit_cluster_s": t_fit,
        "vote_and_write_assignment_csv_s": t_dumped - t_assign,
        "dump_bins_fasta_s": (t_fa1 - t_fa0) if t_fa0 and t_fa1 else None,
        "perform_clustering_total_s": t_total,
    }

    # ---- checks
    got = pd.read_csv(out_csv)
    assert list(got.columns) == ["CONTIG_NAME", "BIN"] and len(got) == n_par
    assert list(got["CONTIG_NAME"]) == sorted(got["CONTIG_NAME"])
    # the vote, independently: most frequent label of each parent, ties to the lowest bin (np.bincount(x).argmax())
    votes = np.zeros((n_par, B), dtype=np.int64)
    np.add.at(votes, (parent_of, labels), 1)
    want_bin = dict(zip(np.char.add("contig", np.arange(n_par).astype(str)), votes.argmax(axis=1)))
    bad = sum(int(want_bin[n] != b) for n, b in zip(got["CONTIG_NAME"], got["BIN"]))
    assert bad == 0, f"{bad} parents with a wrong vote"
    # bins/bin_<i>.fasta: every parent once, in its bin's file
    n_rec = 0
    for b in np.unique(got["BIN"]):
        with open(os.path.join(args.work, "out", "bins", f"bin_{b}.fasta")) as fh:
            ids = [ln[1:].split()[0] for ln in fh if ln.startswith(">")]
        assert all(want_bin[i] == b for i in ids)
        n_rec += len(ids)
    assert n_rec == n_par
    # fixed point of the reference sweep on an oracle sample (all host cores of this box's share)
    try:
        nthr = max(1, min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        nthr = max(1, min(os.cpu_count() or 1, 16))
    sample = rng.choice(np.flatnonzero(initial < 0), args.oracle_sample, replace=False)
    t2 = time.perf_counter()
    bb, _ = O.eval_frozen_mt(X, B, labels, sample, m, nthr)
    t_or = time.perf_counter() - t2
    assert np.array_equal(bb, labels[sample]), "labels are not a fixed point of the reference sweep"

    res = {"config": f"BASELINE configs[4]-shaped: {N} contigs x D={D} ({S} coverage columns) x {B} bins, "
                     f"AlgoNumNeighbors={m}, {args.sub} sub-contigs per parent ({n_par} parents), one MI355X",
           "stages": stages, "generate_inputs_s": t_gen,
           "features_csv_bytes": os.path.getsize(features_csv),
           "accuracy_vs_truth": float((labels == true).mean()),
           "checks": {"assignment_rows": int(len(got)), "vote_mismatches": bad, "fasta_records_in_bins": n_rec,
                      "oracle_fixed_point_sample": int(len(sample)), "oracle_seconds": t_or, "oracle_threads": nthr},
           "log_lines": [msg for _, msg in clock.marks]}
    with open(args.out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res["stages"]), json.dumps(res["checks"]))


if __name__ == "__main__":
    main()
