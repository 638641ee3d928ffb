"""The HIP path on the BASELINE.json configurations at their stated sizes (through the C ABI).

configs[1] (10k x 136 x 32) is small enough for the CPU oracle to replay the whole fit.  configs[3]
(500k x 140 x 128) and configs[4] (1M x 146 x 200) are not: there the result is pinned by
size-independent properties of the reference loop (algorithm.py:43-76):
  (i)   the first contigs of sweep 1 depend only on the seeds and on each other -> exact oracle replay
        of a prefix of the permutation;
  (ii)  a fit that stopped because a sweep changed nothing is a fixed point: every movable contig's label
        is the strict-'>' argmin of its hull distances given everyone else's final label -> checked by the
        oracle (all host cores) on a random sample of contigs;
  (iii) the two-stage selection (fp16 shortlist + exact distances) returns bit-identical member lists to
        the brute-force kernel (CHB_PREFILTER=0) on a sample of contigs under the final labels;
  (iv)  the fused selection + hull-distance kernel and the list-based formulation (CHB_FUSED=0) give the
        same labels for the whole fit, and the speculation statistics stay in the expected range
        (rounds per batch, hull distances evaluated per needed one, shortlist overflows);
  (v)   the call exactly as bench.py makes it (no min_dist: look-ahead across batches with gated kernels and
        host-side snapshot / restore) returns the labels, sweep count and change counts of the call that
        (i)-(iv) pin (want_min_dist=True switches the look-ahead off).
"""
import os

import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_backend import oracle_fit_replay  # noqa: E402

pytestmark = pytest.mark.gpu

QP_TOL = 1e-9


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _ctx(env=None):
    import chbin_amd  # noqa: F401
    from chbin_amd import _lib
    env = env or {}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return _lib.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def _data(N, D, B, S):
    import chbin_amd
    X, initial, true = chbin_amd.synth.make_synthetic(N, D, B, S=S, seed=0)
    perms = chbin_amd.synth.draw_permutations(initial, 4, seed=0)
    return X, initial, true, perms


def _threads():
    try:
        return max(1, min(len(os.sched_getaffinity(0)), 16))
    except AttributeError:
        return max(1, min(os.cpu_count() or 1, 16))


def test_config1_whole_fit_vs_oracle(O):
    """BASELINE configs[1]: 10k contigs x D=136 x 32 bins, m = 5 -- the whole fit against the oracle."""
    N, D, B, m = 10_000, 136, 32, 5
    X, initial, true, perms = _data(N, D, B, 1)
    # (the oracle's whole fit as ONE replay of its sweeps: labels, sweep count, change counts and the last sweep's winning
    #  distances -- tests/oracle_backend.py; `O.fit_cluster` + a second replay for the distances took twice as long)
    want, its_o, ch_o, md = oracle_fit_replay(O, X, B, initial, perms, m, 4)
    c = _ctx()
    try:
        c.set_samples(X)
        assert c.counter("prefilter_enabled") == 1
        got, its, ch, mind = c.fit_cluster(B, initial, perms, m, 4, want_min_dist=True)
        assert c.counter("fused_enabled") == 1
        st = c.fit_stats()
        # the call bench.py times: no min_dist => look-ahead across batches (gated kernels, snapshot / restore)
        got_t, its_t, ch_t = c.fit_cluster(B, initial, perms, m, 4)
    finally:
        c.close()
    assert its == its_o and np.array_equal(ch, ch_o)
    assert np.array_equal(got, want)
    assert its_t == its_o and np.array_equal(ch_t, ch_o) and np.array_equal(got_t, want)
    assert (got == true).mean() > 0.99
    assert st["hull_needed"] == its * perms.shape[1] * B
    # winning distances of the last sweep against the oracle's sequential replay
    assert np.allclose(mind[perms[its - 1]], md, rtol=0, atol=QP_TOL)


def test_config1_function_default_neighbors(O):
    """BASELINE configs[1] with num_neighbors = 15 (algorithm.py:17; list-based path, 16 lanes per hull
    problem): oracle replay of a prefix of sweep 1, then the fixed-point property of the converged fit on
    EVERY movable contig (oracle on all host cores)."""
    N, D, B, m = 10_000, 136, 32, 15
    X, initial, true, perms = _data(N, D, B, 1)
    c = _ctx()
    try:
        c.set_samples(X)
        first, _, _ = c.fit_cluster(B, initial, perms[:1], m, 1)
        got, its, changed, mind = c.fit_cluster(B, initial, perms, m, 4, want_min_dist=True)
    finally:
        c.close()
    n_pref = 300
    lab_o, _ = O.sweep(X, B, initial, perms[0][:n_pref], m)
    assert np.array_equal(first[perms[0][:n_pref]], lab_o[perms[0][:n_pref]])
    assert changed[-1] == 0 and its < 4
    movable = np.flatnonzero(initial < 0)
    bb, bd = O.eval_frozen_mt(X, B, got, movable, m, _threads())
    assert np.array_equal(bb, got[movable])
    assert np.allclose(bd, mind[movable], rtol=0, atol=QP_TOL)


@pytest.mark.parametrize("idx,N,D,B,S", [(3, 500_000, 140, 128, 5), (4, 1_000_000, 146, 200, 10)])
def test_large_configs_full_size(O, idx, N, D, B, S):
    """BASELINE configs[3] and configs[4] at their full sizes on one GPU (see the module docstring)."""
    m = 5
    X, initial, true, perms = _data(N, D, B, S)
    n_move = perms.shape[1]
    c = _ctx()
    try:
        c.set_samples(X)
        assert c.counter("prefilter_enabled") == 1
        # (i) prefix of sweep 1
        first, _, _ = c.fit_cluster(B, initial, perms[:1], m, 1)
        n_pref = 24
        lab_o, _ = O.sweep(X, B, initial, perms[0][:n_pref], m)
        assert np.array_equal(first[perms[0][:n_pref]], lab_o[perms[0][:n_pref]])
        # the whole fit
        got, its, changed, mind = c.fit_cluster(B, initial, perms, m, 4, want_min_dist=True)
        assert c.counter("fused_enabled") == 1
        st = c.fit_stats()
        overflow = c.counter("prefilter_overflow")
        # the call bench.py times (no min_dist => look-ahead across batches): the whole fit once more
        got_t, its_t, changed_t = c.fit_cluster(B, initial, perms, m, 4)
        st_t = c.fit_stats()
        assert c.counter("shortlist_short") == 0   # the product build's check of the shortlist stage's contract
        # (iii) two-stage selection == brute force on a sample of contigs under the final labels
        rng = np.random.default_rng(idx)
        q = rng.choice(np.flatnonzero(initial < 0), 256, replace=False)
        lists = c.topm_per_bin(got, B, m, q)
    finally:
        c.close()
    assert changed[-1] == 0 and its < 4                       # stopped at a fixed point
    # the look-ahead path returns what the path pinned by the oracle properties below returns
    assert its_t == its and np.array_equal(changed_t, changed) and np.array_equal(got_t, got)
    assert st_t["hull_needed"] == st["hull_needed"] and st_t["batches"] == st["batches"]
    assert np.all(got >= 0)
    assert np.array_equal(got[initial >= 0], initial[initial >= 0])   # seeds never move
    assert (got == true).mean() > 0.99
    assert st["hull_needed"] == its * n_move * B
    # (iv) speculation statistics: well-separated bins converge in (almost) one round per batch
    assert st["rounds"] <= 1.5 * st["batches"] + 2
    assert st["hull_evaluated"] <= 1.5 * st["hull_needed"]
    assert overflow <= 1e-4 * its * n_move * B                # the brute-force fallback stays an exception
    # (ii) fixed point, checked by the oracle on a random sample
    sample = rng.choice(np.flatnonzero(initial < 0), 600 if idx == 3 else 400, replace=False)
    bb, bd = O.eval_frozen_mt(X, B, got, sample, m, _threads())
    assert np.array_equal(bb, got[sample])
    assert np.allclose(bd, mind[sample], rtol=0, atol=QP_TOL)
    # (iii)
    b = _ctx({"CHB_PREFILTER": "0"})
    try:
        assert b.counter("prefilter_enabled") == 0
        b.set_samples(X)
        want_lists = b.topm_per_bin(got, B, m, q)
    finally:
        b.close()
    for g, w_ in zip(lists, want_lists):
        assert np.array_equal(g, w_)
    # (iv) fused kernel == list-based formulation, whole fit
    f = _ctx({"CHB_FUSED": "0"})
    try:
        f.set_samples(X)
        got_l, its_l, changed_l = f.fit_cluster(B, initial, perms, m, 4)
        assert f.counter("fused_enabled") == 0
    finally:
        f.close()
    assert its_l == its and np.array_equal(changed_l, changed) and np.array_equal(got_l, got)
