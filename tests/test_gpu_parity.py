"""Parity of the HIP path (through the C ABI) against the CPU oracle and the committed golden
fixtures.  Bit-exact for labels, indices and the cdist-style distances; hull (QP) distances
within 1e-9 absolute (north star tolerance: 1e-5)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

QP_TOL = 1e-9


@pytest.fixture(scope="module")
def ctx():
    import chbin_amd  # noqa: F401
    from chbin_amd import _lib
    return _lib.default_context()


@pytest.fixture(scope="module")
def O():
    from oracle import oracle
    return oracle


def _synth(*a, **k):
    import chbin_amd
    return chbin_amd.synth.make_synthetic(*a, **k)


def _perms(initial, n):
    import chbin_amd
    return chbin_amd.synth.draw_permutations(initial, n, seed=0)


# ------------------------------------------------------------------ distances (K1)

def test_pairwise_bit_exact_vs_golden_and_oracle(ctx, O, golden_dir):
    g = np.load(os.path.join(golden_dir, "cdist.npz"))
    ctx.set_samples(g["X"])
    M = ctx.pairwise_distance()
    assert np.array_equal(M, g["M"])  # scipy cdist, captured from the reference function
    rng = np.random.default_rng(0)
    for N, D in ((300, 136), (257, 140), (65, 7), (1, 3), (1500, 33)):
        X = rng.random((N, D)) / D
        ctx.set_samples(X)
        assert np.array_equal(ctx.pairwise_distance(), O.cdist(X)), (N, D)
    # row ranges
    X = rng.random((500, 136))
    ctx.set_samples(X)
    assert np.array_equal(ctx.pairwise_distance(123, 321), O.cdist(X)[123:321])


def test_mirror_create_distance_matrix(ctx, O, tmp_path):
    from chbin_amd import clustering
    X = np.random.default_rng(1).random((200, 20))
    M = clustering.create_in_mem_distance_matrix(X)
    assert np.array_equal(M, O.cdist(X))
    f = clustering.create_distance_matrix(X, tmp_path)
    assert f.name == "distance_matrix.npy"
    assert np.array_equal(np.load(f), M)
    # reuse-if-exists semantics (distance_matrix.py:19-22)
    assert clustering.create_distance_matrix(X * 2, tmp_path) == f
    assert np.array_equal(np.load(f), M)


# ------------------------------------------------------------------ selection (K2)

def test_find_nearest_from_row_vs_golden(ctx, O, golden_dir):
    from chbin_amd import clustering
    g = np.load(os.path.join(golden_dir, "find_nearest.npz"))
    X, labels, m = g["X"], g["labels"], int(g["m"])
    for i, c, want in zip(g["rows"], g["bins"], g["selected_sorted"]):
        cur = labels.copy()
        cur[i] = -1
        row = O.cdist_row(X, int(i))
        got = clustering.find_nearest_from_cluster(int(c), cur, row, m)
        assert np.array_equal(np.sort(got), want[want >= 0])
        assert np.array_equal(got, O.find_nearest_from_cluster(int(c), cur, row, m))


@pytest.mark.parametrize("m", [1, 5, 15, 16])
def test_topm_per_bin_vs_oracle(ctx, O, m):
    rng = np.random.default_rng(m)
    N, D, B = 900, 136, 7
    X, _, true = _synth(N, D, B, seed=2, sigma=5e-3, mix=0.4)
    X[10] = X[3]; X[11] = X[3]; X[500] = X[499]          # exact duplicates -> distance ties
    labels = true.copy()
    labels[rng.random(N) < 0.25] = -1
    labels[labels == 6] = -1                               # empty bin
    keep = np.flatnonzero(labels == 5)
    labels[keep[3:]] = -1                                  # bin with 3 (< m) members
    labels[[3, 10, 11]] = 2
    ctx.set_samples(X)
    queries = np.concatenate([rng.choice(N, 150, replace=False), [3, 10, 11, 499, 500, 3]])
    idx, dist, cnt = ctx.topm_per_bin(labels, B, m, queries)
    for qi, q in enumerate(queries):
        cur = labels.copy()
        cur[q] = -1
        row = O.cdist_row(X, int(q))
        for c in range(B):
            want = O.find_nearest_from_cluster(c, cur, row, m)
            n = cnt[qi, c]
            assert n == len(want), (q, c)
            assert np.array_equal(idx[qi, c, :n], want), (q, c)
            assert np.array_equal(dist[qi, c, :n], row[want]), (q, c)   # bit-exact distances
            assert np.all(idx[qi, c, n:] == -1)


# ------------------------------------------------------------------ hull distance (K3+K4)

def _hull_cases(rng, n, D, mmax):
    xs, Ps = [], []
    for t in range(n):
        m = int(rng.integers(1, mmax + 1))
        P = rng.random((m, D)) / D
        kind = t % 6
        if kind == 0:
            x = rng.random(D) / D
        elif kind == 1:
            x = rng.dirichlet(np.ones(m)) @ P                     # inside the hull: distance 0
        elif kind == 2:
            x = P[rng.integers(m)] + 1e-4 * rng.standard_normal(D) / D
        elif kind == 3 and m > 1:
            P[m - 1] = P[0]                                       # duplicate vertex (singular Gram)
            x = rng.random(D) / D
        elif kind == 4 and m > 2:
            P[2] = 0.3 * P[0] + 0.7 * P[1]                        # collinear vertices
            x = rng.random(D) / D
        else:
            x = P[0].copy()                                       # query coincides with a vertex
        xs.append(x); Ps.append(P)
    return xs, Ps


@pytest.mark.parametrize("mmax,D", [(5, 136), (8, 136), (16, 140), (5, 3), (12, 6)])
def test_hull_distance_points_vs_oracle_and_enumerator(ctx, O, mmax, D):
    rng = np.random.default_rng(100 + mmax + D)
    xs, Ps = _hull_cases(rng, 120, D, mmax)
    for x, P in zip(xs, Ps):
        d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
        scale = max(np.linalg.norm(P - x, axis=1).max(), 1e-300)
        d_or = O.convex_hull_distance(x, P)
        assert abs(d - d_or) <= QP_TOL + 1e-7 * scale * (d_or < 1e-6 * scale), (len(P), d, d_or)
        if len(P) <= 12:
            d_en = O.enum_hull_distance(x, P)
            assert abs(d - d_en) <= QP_TOL + 1e-7 * scale * (d_en < 1e-6 * scale), (len(P), d, d_en)
        # alpha is a feasible convex weight vector reproducing the distance (hull_distance.py:34-35)
        assert np.all(alpha >= 0) and abs(alpha.sum() - 1) < 1e-12
        assert abs(np.linalg.norm(alpha @ P - x) - d) <= QP_TOL + 1e-7 * scale * (d < 1e-6 * scale)


@pytest.mark.parametrize("D", [3, 7, 40, 136])
def test_hull_distance_16_lane_solver_both_starts(ctx, O, D):
    """The 16-lane solver (5 < m <= 16) starts Wolfe's method from the FULL vertex set when most vertices improve on the
    nearest one (round 5; a query inside its neighbours' cloud) and from the nearest vertex alone otherwise (a query far
    from a tight bin).  Both regimes, affinely dependent sets (m > D + 1: the full-set start has to leave vertices out),
    duplicates and queries inside the hull, m = 6 .. 16, against the oracle's Goldfarb-Idnani and the enumerator."""
    rng = np.random.default_rng(500 + D)
    worst = 0.0
    for t in range(132):
        m = 6 + t % 11
        kind = (t // 11) % 6
        P = rng.standard_normal((m, D))
        if kind == 0:
            x = 0.3 * rng.standard_normal(D)                      # in the cloud: every vertex improves (full-set start)
        elif kind == 1:
            x = 40.0 * np.ones(D) + rng.standard_normal(D)        # far away: support of a few vertices (grown corral)
        elif kind == 2:
            x = rng.dirichlet(np.ones(m)) @ P                     # inside the hull
        elif kind == 3:
            P[m - 1] = P[0]; P[m - 2] = P[1]                      # duplicates in a cloud
            x = 0.3 * rng.standard_normal(D)
        elif kind == 4:
            P[3:] = rng.dirichlet(np.ones(3), size=m - 3) @ P[:3] + 1e-3 * rng.standard_normal((m - 3, D))
            x = P.mean(0) + 0.5 * rng.standard_normal(D)          # nearly coplanar vertices
        else:
            P *= 1e-6; x = 1e-6 * 0.3 * rng.standard_normal(D) + 5.0   # tiny cloud on a large offset
            P += 5.0
        d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
        scale = max(np.linalg.norm(P - x, axis=1).max(), 1e-300)
        d_or = O.convex_hull_distance(x, P)
        if kind == 5:
            # The reference's formulation (G = 2 P P^T on the UNSHIFTED vertices, solve_qp.py:44-51, which the oracle
            # restates) loses the 1e-6 cloud under the offset of 5: it answers 4.5e-8 where the query lies inside the hull
            # (enumerator: 1e-22).  Within the north star's 1e-5 of it; the yardstick is the same problem shifted to the query
            # (the distance does not depend on the origin), where the oracle is well conditioned.
            assert abs(d - d_or) < 1e-5
            d_or = O.convex_hull_distance(np.zeros(D), P - x)
        tol = QP_TOL * max(scale, 1.0) + 1e-7 * scale * (d_or < 1e-6 * scale)
        assert abs(d - d_or) <= tol, (D, m, kind, d, d_or)
        if m <= 12:
            d_en = O.enum_hull_distance(x, P)
            assert abs(d - d_en) <= QP_TOL * max(scale, 1.0) + 1e-7 * scale * (d_en < 1e-6 * scale), (D, m, kind, d, d_en)
        assert np.all(alpha >= 0) and abs(alpha.sum() - 1) < 1e-12
        assert abs(np.linalg.norm(alpha @ P - x) - d) <= tol
        if d_or >= 1e-6 * scale:   # (a distance of zero comes out as the root of a rounding-sized square: ~1e-9 of the scale)
            worst = max(worst, abs(d - d_or) / max(scale, 1e-300))
    assert worst < 1e-9, worst


def test_hull_distance_near_degenerate_sets(ctx, O):
    """Vertex sets that are ALMOST affinely dependent: m = 6 .. 12 points of which all but k <= D lie within a relative
    thickness of 1e-8 .. 1e-1 of the affine hull of the first k (nearly coplanar / collinear neighbours: near-duplicate
    contigs), D = 2 .. 5, against the exhaustive enumerator.  Round 5 found the 16-lane solver's updated inverse losing
    eps * cond^2 when a vertex that had entered on a small pivot left again (a distance 1.2 % off, another 28 % off): it now
    rebuilds the inverse after such a removal.  Envelope: rounding-sized errors (rarely up to eps x cond ~ 1e-8) down to a
    thickness of 1e-5 of the scale;
    thinner sets are treated as dependent somewhere below 1e-6 (pivot < 1e-13), which costs at most that thickness."""
    rngd = np.random.default_rng(7)
    worst_thick, worst_thin = 0.0, 0.0
    for trial in range(900):
        D = int(rngd.integers(2, 6)); m = int(rngd.integers(6, 13))
        P = rngd.standard_normal((m, D))
        k = int(rngd.integers(2, D + 1))
        eps_off = 10.0 ** rngd.uniform(-8, -1)
        P[k:] = rngd.dirichlet(np.ones(k), size=m - k) @ P[:k] + eps_off * rngd.standard_normal((m - k, D))
        x = P.mean(0) + 0.5 * rngd.standard_normal(D)
        d = ctx.hull_distance_points(x, P)
        truth = O.enum_hull_distance(np.zeros(D), P - x)
        scale = np.linalg.norm(P - x, axis=1).max()
        err = abs(d - truth) / scale
        if eps_off >= 1e-5:
            # (rounding-sized -- 1e-15 -- in all but a few cases; the worst seen is 1.2e-8, the per-lane solver of m <= 8 on a
            #  pivot of 6e-9: eps x cond of its from-scratch factorisation)
            worst_thick = max(worst_thick, err)
            assert err <= 2e-7, (trial, D, m, k, eps_off, d, truth)
        else:
            worst_thin = max(worst_thin, err)
            assert err <= 5e-6, (trial, D, m, k, eps_off, d, truth)   # (north star: 1e-5; observed worst 5e-7)
    assert worst_thick < 2e-7 and worst_thin < 5e-6
    # the one-wavefront-per-problem solver of m > 16 keeps its inverse the same way (and got the same remedy); yardstick:
    # Goldfarb-Idnani on the shifted problem, itself good to ~1e-8 on such sets
    for trial in range(60):
        D = int(rngd.integers(3, 6)); m = int(rngd.choice([17, 20, 24, 40]))
        P = rngd.standard_normal((m, D))
        k = int(rngd.integers(2, D + 1))
        eps_off = 10.0 ** rngd.uniform(-5, -1)
        P[k:] = rngd.dirichlet(np.ones(k), size=m - k) @ P[:k] + eps_off * rngd.standard_normal((m - k, D))
        x = P.mean(0) + 0.5 * rngd.standard_normal(D)
        d = ctx.hull_distance_points(x, P)
        truth = O.convex_hull_distance(np.zeros(D), P - x)
        # (eps x cond of the thinnest accepted support: 2.2e-16 / t^2, i.e. 2e-6 at t = 1e-5; seen: 2.3e-7 at t = 2e-5)
        assert abs(d - truth) <= 3e-6 * np.linalg.norm(P - x, axis=1).max(), (trial, D, m, k, eps_off, d, truth)


def test_hull_distance_golden_qp_problems(ctx, O, golden_dir):
    from chbin_amd import clustering
    g = np.load(os.path.join(golden_dir, "qp_args.npz"))
    for k in range(len(g["m"])):
        m = int(g["m"][k])
        x, P = g["x"][k], g["P"][k][:m]
        d = clustering.calculate_distance(x, P, "quadprog", "convex")
        assert abs(d - g["dist_with_oracle_gi"][k]) < QP_TOL
    # ... and against numbers NO solver of this repository produced: the reference's own calculate_distance with scipy's
    # SLSQP answering quadprog.solve_qp (qp_args_slsqp.npz; north-star tolerance 1e-5, observed maximum 2.2e-9)
    s = np.load(os.path.join(golden_dir, "qp_args_slsqp.npz"))
    worst = max(abs(clustering.calculate_distance(g["x"][k], g["P"][k][:int(g["m"][k])], "quadprog", "convex") -
                    s["dist_with_slsqp"][k]) for k in range(len(g["m"])))
    assert worst < 1e-5 and worst < 1e-8, worst


def test_hull_distance_loop_problems_of_the_reference_with_a_second_solver(ctx, golden_dir):
    """504 (contig, bin) evaluations sampled from the reference's own fit_cluster loop on the 600-contig case, with the
    member sets the reference's find_nearest_from_cluster selected and the distances its calculate_distance returned
    while scipy's SLSQP answered quadprog.solve_qp (fit_cluster_flow_slsqp.npz): the indexed hull kernel on the same
    rows.  North-star tolerance 1e-5; observed maximum ~1e-15."""
    f = np.load(os.path.join(golden_dir, "fit_cluster_flow.npz"))
    t = np.load(os.path.join(golden_dir, "fit_cluster_flow_slsqp.npz"))
    ctx.set_samples(f["X"])
    d = ctx.hull_distance_batch(t["loop_query"], t["loop_hull"])
    worst = float(np.abs(d - t["loop_dist_slsqp"]).max())
    assert worst < 1e-5 and worst < 1e-11, worst


def test_hull_distance_batch_indexed(ctx, O):
    rng = np.random.default_rng(7)
    X, _, _ = _synth(400, 136, 4, seed=5, sigma=4e-3)
    ctx.set_samples(X)
    P = 300
    q = rng.integers(0, 400, P)
    hull = np.full((P, 8), -1, dtype=np.int64)
    for p in range(P):
        n = int(rng.integers(0, 9))
        ids = rng.choice(400, n, replace=False)
        slots = np.sort(rng.choice(8, n, replace=False))          # padding anywhere
        hull[p, slots] = ids
    dist, alpha = ctx.hull_distance_batch(q, hull, want_alpha=True)
    for p in range(P):
        ids = hull[p][hull[p] >= 0]
        if len(ids) == 0:
            assert np.isinf(dist[p])
            continue
        want = O.convex_hull_distance(X[q[p]], X[ids])
        assert abs(dist[p] - want) < QP_TOL
        assert np.all(alpha[p][hull[p] < 0] == 0)
        assert abs(np.linalg.norm(alpha[p][hull[p] >= 0] @ X[ids] - X[q[p]]) - dist[p]) < 1e-9


def test_mirror_errors():
    from chbin_amd import clustering
    x = np.zeros(4)
    with pytest.raises(NotImplementedError):
        clustering.calculate_distance(x, np.zeros((2, 4)), "quadprog", "manhattan")
    with pytest.raises(NotImplementedError):
        clustering.fit_cluster(np.zeros((4, 4)), 1, np.zeros(4, dtype=np.int64), None, metric="manhattan")
    with pytest.raises(NotImplementedError):
        clustering.calculate_distance(x, np.zeros((2, 4)), "nosuch", "convex")
    with pytest.raises(NotImplementedError):
        clustering.fit_cluster(np.zeros((4, 4)), 1, np.zeros(4, dtype=np.int64), None, qp_solver="nosuch")


# ------------------------------------------------------------------ the whole loop (a12/a13)

def test_fit_cluster_golden_flow(ctx, O, golden_dir):
    """Labels produced by the reference's own fit_cluster loop (captured fixture)."""
    from chbin_amd import clustering
    g = np.load(os.path.join(golden_dir, "fit_cluster_flow.npz"))
    np.random.seed(0)  # ch_bin.py:22
    got = clustering.fit_cluster(g["X"], int(g["B"]), g["initial"], None,
                                 num_neighbors=int(g["m"]), max_iterations=int(g["max_iter"]))
    assert got.dtype == np.int64
    assert np.array_equal(got, g["labels"])
    # the global RNG was advanced exactly as algorithm.py:45 would have
    after = np.random.random()
    np.random.seed(0)
    pts = np.where(g["initial"] == -1)[0]
    sweeps = O.fit_cluster(g["X"], int(g["B"]), g["initial"], g["perms"], int(g["m"]), int(g["max_iter"]))[1]
    for _ in range(sweeps):
        np.random.permutation(pts)
    assert after == np.random.random()


CASES = [
    # N, D, B, m, max_iter, sigma, mix, n_seed, batch
    (600, 40, 6, 5, 10, 9e-3, 0.5, 10, 0),
    (600, 40, 6, 5, 10, 9e-3, 0.5, 10, 64),
    (600, 40, 6, 5, 10, 9e-3, 0.5, 10, 1),        # batch of one == plain sequential
    (900, 136, 8, 5, 6, 6e-3, 0.6, 8, 257),
    (900, 136, 8, 15, 4, 6e-3, 0.6, 20, 300),     # function-default neighbour count (algorithm.py:17)
    (1200, 140, 5, 5, 10, 9e-3, 0.9, 4, 4096),    # heavy overlap: many movers, many rounds
    (700, 24, 9, 3, 5, 4e-3, 0.0, 2, 128),        # bins start smaller than m
    (2000, 136, 8, 5, 3, 1.5e-3, 0.0, None, 0),   # SURVEY 8(d) generator as-is
    # wide rows (round 5: shadow rows of two to four 144-column slices): k = 5 width on the fused kernel with overlapping
    # bins (update-mode shortlist, repeated rounds), the default neighbour count on the list-based kernels
    (900, 528, 8, 5, 5, 3e-3, 0.6, 8, 257),
    (600, 300, 6, 15, 3, 3e-3, 0.6, 20, 300),
    (1500, 200, 5, 5, 6, 5e-3, 0.9, 4, 2048),
]


@pytest.mark.parametrize("N,D,B,m,iters,sigma,mix,n_seed,batch", CASES)
def test_fit_cluster_labels_bit_exact(ctx, O, N, D, B, m, iters, sigma, mix, n_seed, batch):
    S = 5 if D == 140 else 1
    X, initial, _ = _synth(N, D, B, S=S, seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = _perms(initial, iters)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters)
    ctx.set_samples(X)
    got, its, ch, mind = ctx.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
    assert its == its_o
    assert np.array_equal(ch, ch_o)
    assert np.array_equal(got, want)
    assert np.array_equal(got[initial >= 0], initial[initial >= 0])   # seeds never move
    st = ctx.fit_stats()
    assert st["hull_needed"] == its * perms.shape[1] * B
    # winning distances of the last sweep agree with the oracle's sequential replay
    labels = initial.copy()
    for k in range(its):
        labels, md = O.sweep(X, B, labels, perms[k], m)
    assert np.allclose(mind[perms[its - 1]], md, rtol=0, atol=QP_TOL, equal_nan=True)


def test_fit_cluster_degenerate_inputs(ctx, O):
    # duplicates of whole contigs, a bin id with no seed at all, and a contig equal to a seed
    X, initial, true = _synth(500, 30, 5, seed=9, sigma=5e-3, mix=0.3, n_seed=6)
    X[100:110] = X[0]
    X[200] = X[np.flatnonzero(initial == 1)[0]]
    initial[initial == 4] = -1                      # bin 4 has no members: +inf, never chosen
    perms = _perms(initial, 5)
    want, its_o, _ = O.fit_cluster(X, 5, initial, perms, 5, 5)
    ctx.set_samples(X)
    got, its, _ = ctx.fit_cluster(5, initial, perms, 5, 5, batch=100)
    assert its == its_o and np.array_equal(got, want)
    assert not np.any(got == 4)
    # nothing to move at all
    init2 = true.copy()
    got2, its2, ch2 = ctx.fit_cluster(5, init2, np.zeros((3, 0), dtype=np.int64), 5, 3)
    assert np.array_equal(got2, init2) and its2 == 1 and ch2[0] == 0
    # no seeds at all: every hull is empty, labels stay -1 (cli/clustering.py:79 then raises)
    init3 = np.full(500, -1, dtype=np.int64)
    got3, _, _ = ctx.fit_cluster(5, init3, _perms(init3, 2), 5, 2)
    assert np.all(got3 == -1)


def test_stepwise_two_slices_equals_sequential(ctx, O):
    """The multi-GPU decomposition (each rank evaluates a slice of every batch, labels exchanged
    between rounds) driven from one process with two contexts on the same GPU."""
    from chbin_amd import _lib
    from chbin_amd.distributed import run_sweeps
    X, initial, _ = _synth(800, 64, 6, seed=21, sigma=8e-3, mix=0.5, n_seed=8)
    perms = _perms(initial, 4)
    want, its_o, _ = O.fit_cluster(X, 6, initial, perms, 5, 4)
    c2 = _lib.Context(0)
    try:
        got, its, _ = run_sweeps([ctx, c2], X, 6, initial, perms, 5, 4, batch=200)
    finally:
        c2.close()
    assert its == its_o and np.array_equal(got, want)


@pytest.mark.parametrize("N,D,B,S", [(4000, 136, 16, 1)])
def test_config2_scale_parity(ctx, O, N, D, B, S):
    """A BASELINE config-2 shaped case at a size the oracle finishes in seconds."""
    X, initial, true = _synth(N, D, B, S=S, seed=0)
    perms = _perms(initial, 3)
    want, its_o, _ = O.fit_cluster(X, B, initial, perms, 5, 3)
    ctx.set_samples(X)
    got, its, _ = ctx.fit_cluster(B, initial, perms, 5, 3)
    assert its == its_o and np.array_equal(got, want)
    assert (got == true).mean() > 0.99


def test_full_size_fixed_point_property(ctx, O):
    """BASELINE config 3 (N=100k, D=136, B=64, m=5).  The oracle cannot replay 6.4M QPs per sweep,
    but two size-independent properties pin the result: (1) the first contigs of sweep 1 depend only
    on the seeds and on each other, so the oracle replays that prefix exactly; (2) once a sweep
    changes nothing, every contig's label is the argmin of its hull distances given everyone
    else's final label -- checked by the oracle on a random sample."""
    N, D, B, m = 100_000, 136, 64, 5
    X, initial, true = _synth(N, D, B, seed=0)
    perms = _perms(initial, 4)
    ctx.set_samples(X)
    first, _, _ = ctx.fit_cluster(B, initial, perms[:1], m, 1)
    n_pref = 40
    lab_o, _ = O.sweep(X, B, initial, perms[0][:n_pref], m)
    assert np.array_equal(first[perms[0][:n_pref]], lab_o[perms[0][:n_pref]])
    got, its, changed, mind = ctx.fit_cluster(B, initial, perms, m, 4, want_min_dist=True)
    assert changed[-1] == 0 and its < 4
    assert np.all(got >= 0)
    assert np.array_equal(got[initial >= 0], initial[initial >= 0])
    rng = np.random.default_rng(3)
    sample = rng.choice(np.flatnonzero(initial < 0), 30, replace=False)
    for j in sample:
        lab_j, md = O.sweep(X, B, got, np.array([j]), m)
        assert lab_j[j] == got[j]
        assert abs(md[0] - mind[j]) < QP_TOL
    # the call exactly as bench.py times it (no min_dist => look-ahead across batches: gated kernels, host-side
    # snapshot / restore) returns the fit pinned above; the first sweep alone equals sweep 1 of that fit
    got_t, its_t, changed_t = ctx.fit_cluster(B, initial, perms, m, 4)
    st_t = ctx.fit_stats()
    assert its_t == its and np.array_equal(changed_t, changed) and np.array_equal(got_t, got)
    assert st_t["batches"] >= 12 * its                        # every sweep ran its 12-13 gated batches
    one, _, ch1, mind1 = ctx.fit_cluster(B, initial, perms[:1], m, 1, want_min_dist=True)
    assert np.array_equal(first, one) and ch1[0] == changed[0]


def _spec_off_ctx():
    from chbin_amd import _lib
    old = os.environ.get("CHB_SPECULATE")
    os.environ["CHB_SPECULATE"] = "0"
    try:
        return _lib.Context(0)
    finally:
        if old is None:
            del os.environ["CHB_SPECULATE"]
        else:
            os.environ["CHB_SPECULATE"] = old


def test_lookahead_on_overlapping_bins_full_size(ctx, O):
    """N=100k on overlapping bins (mix 0.3 / sigma 4.5e-3: ~3.5 rounds per batch, so the look-ahead of
    chb_fit_cluster keeps failing, restoring the host-side batch state and re-enqueueing the next batch).  The
    default context must return exactly what a context without look-ahead (CHB_SPECULATE=0) returns, and the
    oracle replays a prefix of sweep 1 and checks the fixed-point property of sweep 2's visits on a sample."""
    N, D, B, m = 100_000, 136, 64, 5
    X, initial, true = _synth(N, D, B, seed=0, mix=0.3, sigma=4.5e-3)
    perms = _perms(initial, 2)
    ctx.set_samples(X)
    got, its, changed = ctx.fit_cluster(B, initial, perms, m, 2)
    st = ctx.fit_stats()
    assert st["rounds"] > 1.5 * st["batches"]                 # rounds did repeat: the restore path ran
    plain = _spec_off_ctx()
    try:
        plain.set_samples(X)
        want, its_w, changed_w, mind = plain.fit_cluster(B, initial, perms, m, 2, want_min_dist=True)
        st_w = plain.fit_stats()
    finally:
        plain.close()
    assert its == its_w and np.array_equal(changed, changed_w) and np.array_equal(got, want)
    assert st["hull_needed"] == st_w["hull_needed"]
    n_pref = 40
    first, _, _ = ctx.fit_cluster(B, initial, perms[:1], m, 1)
    lab_o, _ = O.sweep(X, B, initial, perms[0][:n_pref], m)
    assert np.array_equal(first[perms[0][:n_pref]], lab_o[perms[0][:n_pref]])
    # the LAST contigs of the last sweep were visited when every other label was already final
    tail = perms[its - 1][-24:]
    for k, j in enumerate(tail):
        lab_now = got.copy()
        lab_now[tail[k:]] = first[tail[k:]] if its == 2 else initial[tail[k:]]   # labels at j's visit
        lab_j, md = O.sweep(X, B, lab_now, np.array([j]), m)
        assert lab_j[j] == got[j]
        assert abs(md[0] - mind[j]) < QP_TOL


# ------------------------------------------------------------------ two-stage selection

def _brute_ctx():
    """A second context with the fp16 shortlist stage disabled (brute-force selection)."""
    from chbin_amd import _lib
    old = os.environ.get("CHB_PREFILTER")
    os.environ["CHB_PREFILTER"] = "0"
    try:
        c = _lib.Context(0)
    finally:
        if old is None:
            del os.environ["CHB_PREFILTER"]
        else:
            os.environ["CHB_PREFILTER"] = old
    return c


@pytest.mark.parametrize("kind", ["offset", "tiny", "dups", "wide", "onehot", "bigbins", "d141", "d157", "spread",
                                  "d158", "d300", "d528", "d573", "d528offset", "d300dups", "d285bigbins", "d573spread"])
def test_shortlist_stage_equals_brute_force(ctx, O, kind):
    """The fp16 shortlist + exact rescoring must give bit-identical lists to the brute-force
    kernel on data built to stress the error bounds and the overflow fallback."""
    rng = np.random.default_rng(11)
    N, D, B, m = 3000, 136, 6, 5
    if kind == "bigbins":
        N, B = 9000, 4          # > 512 members per bin: the per-tile-best threshold mode of sweep 0
    if kind == "d141":
        D = 141                 # the widest rows of the 144-column build: exactly three spare (bias) columns
    if kind == "d157":
        D = 157                 # ... of the 160-column build
    # wide rows (round 5): two to four 144-column slices per shadow row -- the narrowest (158 = two slices), k = 5 (512
    # k-mer columns + 16), the widest (573 = 4 x 144 - 3), with the stress kinds of the narrow builds on top
    if kind.startswith("d158"):
        D = 158
    if kind.startswith("d300"):
        D = 300
    if kind.startswith("d528"):
        D = 528
    if kind.startswith("d573"):
        D = 573
    if kind == "d285bigbins":
        D, N, B = 285, 9000, 4  # exactly two slices; > 512 members per bin (per-tile bests)
    X, _, true = _synth(N, D, B, seed=4, sigma=3e-3, mix=0.5)
    if kind in ("offset", "d528offset"):
        X = X + 1000.0                                   # huge common offset: centring must cope
    elif kind == "tiny":
        X = X * 1e-150
    elif kind in ("dups", "d300dups"):
        X[100:400] = X[100]                              # 300 identical members: shortlist overflow
        true[100:400] = 1
    elif kind == "wide":
        X = X * rng.lognormal(0, 3, size=(1, D))         # wildly different column scales
    elif kind == "onehot":
        X = np.zeros((N, D)); X[np.arange(N), rng.integers(0, D, N)] = 1.0   # massive exact ties
    elif kind in ("spread", "d573spread"):
        # bins 2^17 times further apart than they are wide: the members' bias (carried as three fp16 pieces in the
        # shadow rows) is huge against the distances that decide the selection
        X = X + 1e3 * true[:, None] * rng.random((1, D))
    labels = true.copy()
    labels[rng.random(N) < 0.1] = -1
    queries = rng.choice(N, 700, replace=False)
    ctx.set_samples(X)
    assert ctx.counter("prefilter_enabled") == 1   # (known once the samples are in: the shadow rows must fit)
    got = ctx.topm_per_bin(labels, B, m, queries)
    overflow = ctx.counter("prefilter_overflow")
    b = _brute_ctx()
    try:
        assert b.counter("prefilter_enabled") == 0
        b.set_samples(X)
        want = b.topm_per_bin(labels, B, m, queries)
    finally:
        b.close()
    for g, w_ in zip(got, want):
        assert np.array_equal(g, w_)
    if kind in ("dups", "onehot", "d300dups"):
        assert overflow > 0                              # the fallback really ran
    # and against the oracle for a few queries
    for qi in range(0, 700, 97):
        q = queries[qi]
        cur = labels.copy(); cur[q] = -1
        row = O.cdist_row(X, int(q))
        for c in range(B):
            want_idx = O.find_nearest_from_cluster(c, cur, row, m)
            assert np.array_equal(got[0][qi, c, :len(want_idx)], want_idx)


@pytest.mark.parametrize("N,D,B,m", [(20000, 136, 500, 5), (12000, 140, 64, 15), (9000, 146, 3, 8), (12000, 136, 12, 15),
                                     (16000, 528, 12, 5), (9000, 300, 6, 15), (9000, 573, 5, 8)])
def test_fit_two_stage_equals_brute_force_selection(O, N, D, B, m):
    """Whole fits with the two-stage selection against the same fits with CHB_PREFILTER=0 (brute-force
    selection kernel) at sizes the oracle cannot replay: many small bins, long lists (m = 15, where
    the shortlist pool is emptied mid-bin and overflow fallbacks occur), few huge bins, and (round 4) m = 15 on bins
    of ~1000 members = 31 tiles: >= 16 tiles but 4 x tiles < m^2, where sweep 0 lets the three best values per tile
    half compete; (round 5) wide rows -- k = 5 (528 columns, four slices) with the persistent pack and the fused m <= 5
    kernel, three slices with m = 15 on the list-based kernels, the widest rows (573) with m = 8."""
    from chbin_amd import _lib
    S = 1 if (D <= 136 or D > 160) else (5 if D == 140 else 10)
    X, initial, _ = _synth(N, D, B, S=S, seed=3, sigma=3e-3, mix=0.3, n_seed=3)
    perms = _perms(initial, 3)
    a = _lib.Context(0)
    try:
        a.set_samples(X)
        assert a.counter("prefilter_enabled") == 1
        la, ia, ca = a.fit_cluster(B, initial, perms, m, 3)
    finally:
        a.close()
    b = _brute_ctx()
    try:
        b.set_samples(X)
        assert b.counter("prefilter_enabled") == 0
        lb, ib, cb = b.fit_cluster(B, initial, perms, m, 3)
    finally:
        b.close()
    assert ia == ib and np.array_equal(ca, cb) and np.array_equal(la, lb)


# ------------------------------------------------------------------ clustering-stage driver (8f-1)

def test_cli_perform_clustering_matches_oracle(ctx, O, tmp_path, golden_dir, monkeypatch):
    """BASELINE configs[0] plumbing: the reference's own contig names / coverages
    (five-genomes-abundance.abund) + synthetic k-mer profiles -> features.csv -> HIP fit ->
    binning-assignment.csv, identical to the same driver with the oracle's fit injected."""
    import sys
    import pandas as pd
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from test_cli_cpu import make_features_csv, oracle_fit
    from chbin_amd import cli_clustering
    feats = tmp_path / "features.csv"
    make_features_csv(feats, golden_dir)
    # a contig FASTA for the parents of that table (plus one record nobody is assigned to), so that the GPU run goes
    # through dump_bins (dump_bins.py:8-29, called at cli/clustering.py:96) as the reference's does
    parents = sorted(set(pd.read_csv(feats)["PARENT_NAME"].astype(str)))
    fasta = tmp_path / "contigs.fasta"
    rng = np.random.default_rng(3)
    with open(fasta, "w") as fh:
        for name in parents + ["not_binned_contig"]:
            fh.write(f">{name} len=130\n")
            seq = "".join(rng.choice(list("ACGT"), 130))
            fh.write(seq[:70] + "\n" + seq[70:] + "\n")
    np.random.seed(0)
    out_gpu = cli_clustering.perform_clustering(fasta, feats, tmp_path / "gpu", num_neighbors=5,
                                                max_iterations=6)
    monkeypatch.setattr(cli_clustering, "fit_cluster", oracle_fit)
    np.random.seed(0)
    out_cpu = cli_clustering.perform_clustering(fasta, feats, tmp_path / "cpu", num_neighbors=5,
                                                max_iterations=6)
    assert open(out_gpu).read() == open(out_cpu).read()
    table = pd.read_csv(out_gpu)
    assert len(table) == 120
    # bins/bin_<i>.fasta: every assigned parent exactly once, in its bin's file, header line kept; the unassigned
    # record dropped; the two runs' files identical
    seen = {}
    for b in sorted(set(table["BIN"])):
        text = open(tmp_path / "gpu" / "bins" / f"bin_{b}.fasta").read()
        assert text == open(tmp_path / "cpu" / "bins" / f"bin_{b}.fasta").read()
        for line in text.splitlines():
            if line.startswith(">"):
                assert line.endswith(" len=130")
                seen[line[1:].split()[0]] = b
    assert seen == dict(zip(table["CONTIG_NAME"].astype(str), table["BIN"]))


def test_native_comm_exchange_path_world1(O):
    """The RCCL exchange path of chb_fit_cluster (communicator of one rank, all-gathers forced on), with the samples
    set through chb_bcast_samples: same labels and winning distances as the plain path."""
    from chbin_amd import _lib
    X, initial, _ = _synth(900, 136, 8, seed=31, sigma=6e-3, mix=0.6, n_seed=8)
    perms = _perms(initial, 4)
    want, its_o, _ = O.fit_cluster(X, 8, initial, perms, 5, 4)
    old = os.environ.get("CHB_FORCE_GATHER")
    os.environ["CHB_FORCE_GATHER"] = "1"
    try:
        c = _lib.Context(0)
    finally:
        if old is None:
            del os.environ["CHB_FORCE_GATHER"]
        else:
            os.environ["CHB_FORCE_GATHER"] = old
    try:
        with pytest.raises(_lib.ChbError, match="chb_comm_init"):
            c.bcast_samples(X, X.shape[0], X.shape[1])       # no communicator yet
        assert c.comm_info() == {"rank": 0, "world": 1, "comm_ranks": 0, "transport": "none"}
        c.comm_init(_lib.Context.comm_unique_id(), 0, 1)
        assert c.comm_info() == {"rank": 0, "world": 1, "comm_ranks": 1, "transport": "rccl"}
        c.bcast_samples(X, X.shape[0], X.shape[1], root=0)   # the library's RCCL broadcast (one rank: a copy onto itself)
        got, its, _, mind = c.fit_cluster(8, initial, perms, 5, 4, batch=200, want_min_dist=True)
        assert its == its_o and np.array_equal(got, want)
        labels = initial.copy()
        for k in range(its):
            labels, md = O.sweep(X, 8, labels, perms[k], 5)
        assert np.allclose(mind[perms[its - 1]], md, rtol=0, atol=QP_TOL)
        # look-ahead UNDER the exchange (no min_dist => the next batch is enqueued behind the all-gather and the device-side
        # first-change reduction, gated on their verdict): well separated bins converge in one round per batch, so the
        # look-ahead holds; the overlapping case above makes it fail and restore.  Both must equal the plain path.
        X2, initial2, _ = _synth(5000, 136, 16, seed=33)
        perms2 = _perms(initial2, 3)
        want2, its2, ch2 = O.fit_cluster(X2, 16, initial2, perms2, 5, 3)
        c.bcast_samples(X2, X2.shape[0], X2.shape[1], root=0)
        got2, its_g, ch_g = c.fit_cluster(16, initial2, perms2, 5, 3, batch=256)
        assert c.counter("lookahead_batches") > 0, "the look-ahead never engaged under the RCCL exchange"
        assert its_g == its2 and np.array_equal(ch_g, ch2) and np.array_equal(got2, want2)
        c.bcast_samples(X, X.shape[0], X.shape[1], root=0)
        got3, its3, _ = c.fit_cluster(8, initial, perms, 5, 4, batch=200)      # failing look-aheads, restores
        assert its3 == its_o and np.array_equal(got3, want)
        c.comm_destroy()
    finally:
        c.close()


@pytest.mark.parametrize("N,D,B,m", [(400, 300, 3, 5), (300, 8, 1, 4), (900, 146, 40, 5), (64, 16, 2, 16),
                                     (700, 200, 4, 15), (500, 528, 3, 5), (300, 573, 2, 8), (300, 574, 2, 5)])
def test_fit_cluster_odd_shapes(ctx, O, N, D, B, m):
    """157 < D <= 573 (the wide shortlist builds, round 5; the fused 16-lane kernel sweeps rows beyond 288 columns in
    windows), D = 574 (no fp16 shadow: brute-force selection), a single bin, many bins with few members
    each, D = 146 (10 coverage columns: 10 MFMA k-steps), m = 16 with bins smaller than m."""
    S = 10 if D == 146 else 1
    X, initial, _ = _synth(N, D, B, S=S, seed=D + B, sigma=6e-3, mix=0.5, n_seed=3)
    perms = _perms(initial, 4)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, 4)
    ctx.set_samples(X)
    assert ctx.counter("prefilter_enabled") == (1 if D <= 573 else 0)
    got, its, ch = ctx.fit_cluster(B, initial, perms, m, 4, batch=150)
    assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    if D > 160:
        # (both fused kernels take rows of any width: the 16-lane one stages its query row in LDS in windows of 288 columns)
        assert ctx.counter("fused_enabled") == (1 if D <= 573 else 0)


def test_hull_distance_kkt_properties(ctx, O):
    """Solver-independent optimality check of the GPU hull weights: feasibility, stationarity on
    the support, dual feasibility off it, and invariance under vertex permutation / translation."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=400, deadline=None, derandomize=True)
    @given(st.integers(1, 16), st.integers(2, 140), st.integers(0, 2**31 - 1), st.floats(1e-4, 10.0))
    def check(m, D, seed, scale):
        rng = np.random.default_rng(seed)
        P = scale * rng.random((m, D)) / D
        x = scale * rng.random(D) / D
        if seed % 3 == 0 and m > 1:
            P[-1] = P[0]
        d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
        Y = P - x
        Q = Y @ Y.T
        g = Q @ alpha
        val = float(alpha @ g)
        tol = 1e-9 * max(np.diag(Q).max(), 1e-300)
        assert np.all(alpha >= 0) and abs(alpha.sum() - 1) < 1e-12
        assert abs(d * d - val) <= 10 * tol + 1e-12 * val
        assert np.all(g >= val - 1e3 * tol)                      # no vertex improves (dual feasibility)
        assert np.all(np.abs(g[alpha > 1e-9] - val) <= 1e3 * tol)  # stationarity on the support
        # a query inside the hull returns sqrt(residual of size ~64 eps max Q_ii), not exactly 0
        near0 = 2e-7 * np.sqrt(np.diag(Q).max())
        perm = rng.permutation(m)
        assert abs(ctx.hull_distance_points(x, P[perm]) - d) <= 1e-9 * scale + near0
        shift = scale * rng.random(D)
        assert abs(ctx.hull_distance_points(x + shift, P + shift) - d) <= 1e-8 * scale + near0

    check()


# ------------------------------------------------------------------ affine metrics (8f-4)

def test_affine_hull_distance_vs_reference_formula(ctx, O):
    """hull_distance.py:69-87 restated with numpy/scipy (orth basis, projector) and the oracle's
    Gram-Schmidt version, incl. collinear / duplicate vertices and more vertices than dimensions."""
    import scipy.linalg
    from chbin_amd import clustering

    def ref(q, P):
        mean = P.mean(axis=0)
        basis = scipy.linalg.orth((P - mean).T)
        if basis.shape[1] == 0:
            return np.linalg.norm(q - mean)
        proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
        return np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (q - mean))

    rng = np.random.default_rng(17)
    for t in range(150):
        m = int(rng.integers(1, 17))
        D = int(rng.integers(2, 140))
        P = rng.random((m, D)) / D
        x = rng.random(D) / D
        if t % 4 == 1 and m > 2:
            P[2] = 0.3 * P[0] + 0.7 * P[1]
        if t % 4 == 2 and m > 1:
            P[-1] = P[0]
        want = ref(x, P)
        scale = np.linalg.norm(P - x, axis=1).max()
        for metric in ("affine", "affine-qp"):
            got = clustering.calculate_distance(x, P, "quadprog", metric)
            assert abs(got - want) <= 1e-9 + 2e-7 * scale * (want < 1e-6 * scale), (t, m, D, got, want)
        assert abs(O.affine_hull_distance(x, P) - want) < 1e-12
        # the affine hull contains the convex hull
        assert clustering.calculate_distance(x, P, "quadprog", "convex") >= want - 1e-9


@pytest.mark.parametrize("metric", ["affine", "affine-qp"])
def test_fit_cluster_affine_metric(ctx, O, metric):
    from chbin_amd import clustering
    X, initial, _ = _synth(800, 64, 6, seed=5, sigma=8e-3, mix=0.5, n_seed=8)
    perms = _perms(initial, 5)
    want, its_o, _ = O.fit_cluster(X, 6, initial, perms, 5, 5, metric=metric)
    np.random.seed(0)
    got = clustering.fit_cluster(X, 6, initial, None, num_neighbors=5, max_iterations=5, metric=metric)
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ the solve_qp seam (a8 / a11)

def test_solve_qp_reference_forms(ctx, O, golden_dir):
    """solve_qp.py:96-132 with the two argument tuples hull_distance.py builds (:17-33 simplex form, :48-64
    equality only), for every solver name the reference accepts.  The weights reproduce the hull distance of
    the oracle's Goldfarb-Idnani solve of the quadprog tuple CAPTURED from the reference (qp_args.npz)."""
    from chbin_amd import clustering
    g = np.load(os.path.join(golden_dir, "qp_args.npz"))
    for k in range(len(g["m"])):
        m = int(g["m"][k])
        P, x = g["P"][k][:m], g["x"][k]
        mat_p, vec_q = 2.0 * P @ P.T, -2.0 * P @ x              # hull_distance.py:30-31
        args = (mat_p, vec_q, -np.eye(m), np.zeros(m), np.ones((1, m)), np.ones(1))
        alpha_gi = O.gi_solve(g["G"][k][:m, :m], g["a"][k][:m], g["C"][k][:m, : m + 1], g["b"][k][: m + 1], 1)
        d_gi = np.linalg.norm(alpha_gi @ P - x)
        for solver in ("quadprog", "cvxopt", "hip"):
            alpha = clustering.solve_qp(*args, solver=solver)
            assert alpha.shape == (m,) and np.all(alpha >= 0) and abs(alpha.sum() - 1) < 1e-12
            assert abs(np.linalg.norm(alpha @ P - x) - d_gi) < 1e-9
            assert abs(np.linalg.norm(alpha @ P - x) - g["dist_with_oracle_gi"][k]) < 1e-9
        # equality-only form (affine hull, hull_distance.py:48-64)
        # ... with the tuple hull_distance.py:45-55 really builds (a ZERO-ROW inequality block), and with None / None
        from chbin_amd._lib import default_context
        default_context().set_metric("affine")       # a caller's metric must survive the call
        for gz, hz in ((np.zeros(shape=(0, m)), np.zeros(0)), (None, None)):
            beta = clustering.solve_qp(mat_p, vec_q, gz, hz, np.ones((1, m)), np.ones(1), solver="quadprog")
            assert abs(beta.sum() - 1) < 1e-10
            assert abs(np.linalg.norm(beta @ P - x) - O.affine_hull_distance(x, P)) < 1e-9
        assert default_context().get_metric() == "affine"
        default_context().set_metric("convex")
    with pytest.raises(NotImplementedError):
        clustering.solve_qp(np.eye(3), np.zeros(3), -np.eye(3), np.zeros(3), np.ones((1, 3)), np.ones(1), solver="nosuch")
    with pytest.raises(NotImplementedError):   # not a form the reference poses
        clustering.solve_qp(np.eye(3), np.zeros(3), np.eye(3), np.ones(3), np.ones((1, 3)), np.ones(1))


def test_cvxopt_solver_name_runs_the_same_path(ctx, O):
    """AlgoQpSolver=cvxopt (solve_qp.py:54-93, :129): same distances, same labels as quadprog."""
    from chbin_amd import clustering
    rng = np.random.default_rng(5)
    xs, Ps = _hull_cases(rng, 40, 136, 8)
    for x, P in zip(xs, Ps):
        d_c = clustering.calculate_distance(x, P, "cvxopt", "convex")
        assert d_c == clustering.calculate_distance(x, P, "quadprog", "convex")
        assert abs(d_c - O.convex_hull_distance(x, P)) <= QP_TOL + 1e-7 * np.linalg.norm(P - x, axis=1).max() * (d_c < 1e-8)
    X, initial, _ = _synth(700, 64, 6, seed=12, sigma=8e-3, mix=0.5, n_seed=8)
    perms = _perms(initial, 4)
    want, _, _ = O.fit_cluster(X, 6, initial, perms, 5, 4)
    np.random.seed(0)
    got = clustering.fit_cluster(X, 6, initial, None, num_neighbors=5, max_iterations=4, qp_solver="cvxopt")
    assert np.array_equal(got, want)


# ------------------------------------------------------------------ num_neighbors > 16 (no cap in the reference)

@pytest.mark.parametrize("m", [17, 24, 40, 64])
def test_hull_distance_more_than_16_vertices(ctx, O, m):
    rng = np.random.default_rng(m)
    xs, Ps = _hull_cases(rng, 36, 136 if m < 64 else 70, m)
    for x, P in zip(xs, Ps):
        if len(P) <= 16:
            P = np.vstack([P, rng.random((17 - len(P), P.shape[1])) / P.shape[1]])
        d, alpha = ctx.hull_distance_points(x, P, want_alpha=True)
        scale = max(np.linalg.norm(P - x, axis=1).max(), 1e-300)
        d_or = O.convex_hull_distance(x, P)
        assert abs(d - d_or) <= QP_TOL + 1e-7 * scale * (d_or < 1e-6 * scale), (len(P), d, d_or)
        assert np.all(alpha >= 0) and abs(alpha.sum() - 1) < 1e-12
        assert abs(np.linalg.norm(alpha @ P - x) - d) <= QP_TOL + 1e-7 * scale * (d < 1e-6 * scale)


@pytest.mark.parametrize("N,D,B,m,batch", [(500, 40, 3, 20, 0), (400, 136, 2, 33, 60), (220, 24, 2, 64, 0)])
def test_fit_cluster_more_than_16_neighbors(ctx, O, N, D, B, m, batch):
    """AlgoNumNeighbors beyond the tuned kernels' 16: plain kernels, same labels as the oracle."""
    X, initial, _ = _synth(N, D, B, seed=N + m, sigma=6e-3, mix=0.5, n_seed=m + 6)
    perms = _perms(initial, 3)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, 3)
    ctx.set_samples(X)
    got, its, ch, mind = ctx.fit_cluster(B, initial, perms, m, 3, batch=batch, want_min_dist=True)
    assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    labels = initial.copy()
    for k in range(its):
        labels, md = O.sweep(X, B, labels, perms[k], m)
    assert np.allclose(mind[perms[its - 1]], md, rtol=0, atol=QP_TOL)
    # the per-bin selection on its own, lists of more than 16
    rng = np.random.default_rng(1)
    q = rng.choice(N, 40, replace=False)
    idx, dist, cnt = ctx.topm_per_bin(got, B, m, q)
    for qi, qq in enumerate(q):
        cur = got.copy(); cur[qq] = -1
        row = O.cdist_row(X, int(qq))
        for c in range(B):
            wl = O.find_nearest_from_cluster(c, cur, row, m)
            assert cnt[qi, c] == len(wl) and np.array_equal(idx[qi, c, :len(wl)], wl)
            assert np.array_equal(dist[qi, c, :len(wl)], row[wl])


# ------------------------------------------------------------------ regressions (round-1 review)

def test_mirror_fit_cluster_reuploads_samples(ctx, O):
    """Two fits through the reference-signature mirror with DIFFERENT Fortran-ordered / float32 arrays of the
    same shape: each must be clustered on its own data (the mirror used to skip the upload when the
    temporary C-contiguous copy landed at the previous call's address)."""
    from chbin_amd import clustering
    wants, gots = [], []
    for seed, dtype in ((3, np.float64), (4, np.float64), (5, np.float32), (6, np.float32)):
        X, initial, _ = _synth(1200, 136, 6, seed=seed, sigma=6e-3, mix=0.5, n_seed=8)
        Xf = np.asfortranarray(X.astype(dtype))                # what DataFrame.values may hand over
        perms = _perms(initial, 3)
        wants.append(O.fit_cluster(np.ascontiguousarray(Xf, dtype=np.float64), 6, initial, perms, 5, 3)[0])
        np.random.seed(0)
        gots.append(clustering.fit_cluster(Xf, 6, initial, None, num_neighbors=5, max_iterations=3))
    for w_, g_ in zip(wants, gots):
        assert np.array_equal(g_, w_)
    assert not np.array_equal(wants[0], wants[1])              # (the cases really differ)
    # in-place modification between two calls
    X, initial, _ = _synth(1200, 136, 6, seed=9, sigma=6e-3, mix=0.5, n_seed=8)
    perms = _perms(initial, 3)
    np.random.seed(0)
    first = clustering.fit_cluster(X, 6, initial, None, num_neighbors=5, max_iterations=3)
    X[:] = _synth(1200, 136, 6, seed=10, sigma=6e-3, mix=0.5, n_seed=8)[0]
    np.random.seed(0)
    second = clustering.fit_cluster(X, 6, initial, None, num_neighbors=5, max_iterations=3)
    assert np.array_equal(second, O.fit_cluster(X, 6, initial, perms, 5, 3)[0])
    assert not np.array_equal(first, second)


@pytest.mark.parametrize("batch,n_seed", [(32, 1), (2, 1), (63, 2), (17, 3)])
def test_small_batch_fewer_seeds_than_batch(ctx, O, batch, n_seed):
    """A caller batch below 64 with fewer labelled members than the batch: the sweep-1 batch schedule must
    stay inside the buffers sized for `batch`."""
    X, initial, _ = _synth(500, 40, 6, seed=batch, sigma=8e-3, mix=0.4, n_seed=n_seed)
    perms = _perms(initial, 3)
    want, its_o, ch_o = O.fit_cluster(X, 6, initial, perms, 5, 3)
    ctx.set_samples(X)
    got, its, ch = ctx.fit_cluster(6, initial, perms, 5, 3, batch=batch)
    assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    from chbin_amd.distributed import batch_schedule
    assert all(K <= batch for _, K in batch_schedule(perms.shape[1], batch, int((initial >= 0).sum()), True))


# ------------------------------------------------------------------ fused selection + hull kernel, 5 < m <= 16

F16_CASES = [
    # N, D, B, m, n_seed, batch, dup  (sigma 6e-3, mix 0.5: overlapping bins, several rounds per batch)
    (1000, 136, 5, 15, 20, 0, False),      # one matrix-core tile (+ 1..2 extra rows) per pair
    (1000, 136, 5, 16, 20, 0, False),      # m = 16: every tile row is a vertex
    (1500, 136, 6, 6, 8, 0, False),        # short lists on the 16-lane kernel
    (1200, 140, 5, 9, 12, 600, False),
    (1200, 40, 4, 12, 3, 800, False),      # few seeds, large batches: most candidates are batch entries (two tiles, exact path)
    (800, 157, 3, 15, 20, 0, False),       # the longest rows the shortlist stage takes (157 + 3 bias columns = 160)
    (1000, 64, 4, 15, 20, 0, True),        # duplicated members: exact ties at the selection boundary -> exact path
    # rows wider than the 288-column tile that stages the query row (round 5): the WIDEROW instantiation's windows --
    # two windows with a short second one, k = 5 width (two windows), the widest rows (576 = exactly two), many batch entries
    (600, 300, 4, 15, 20, 0, False),
    (600, 528, 4, 15, 20, 0, False),
    (700, 573, 3, 16, 20, 0, False),
    (900, 300, 4, 12, 3, 600, False),
    (500, 528, 4, 15, 20, 0, True),
]


@pytest.mark.parametrize("N,D,B,m,n_seed,batch,dup", F16_CASES)
def test_fused_16_lane_kernel_vs_oracle_and_lists(O, N, D, B, m, n_seed, batch, dup):
    """The fused kernel for 5 < m <= 16 (qp_kernels.hip: hull_select_qp16_kernel) against the oracle and against
    the list-based formulation (CHB_FUSED=0) of the same library: labels, sweeps and winning distances."""
    from chbin_amd import _lib
    S = 5 if D == 140 else 1
    X, initial, _ = _synth(N, D, B, S=S, seed=N + m, sigma=6e-3, mix=0.5, n_seed=n_seed)
    if dup:
        rng = np.random.default_rng(1)
        src = rng.choice(N, N // 3, replace=False)
        dst = rng.choice(N, N // 3, replace=False)
        X[dst] = X[src]
    perms = _perms(initial, 3)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, 3)
    a = _lib.Context(0)
    try:
        a.set_samples(X)
        got, its, ch, mind = a.fit_cluster(B, initial, perms, m, 3, batch=batch, want_min_dist=True)
        assert a.counter("fused_enabled") == 1
    finally:
        a.close()
    assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    labels = initial.copy()
    for k in range(its):
        labels, md = O.sweep(X, B, labels, perms[k], m)
    assert np.allclose(mind[perms[its - 1]], md, rtol=0, atol=QP_TOL, equal_nan=True)
    old = os.environ.get("CHB_FUSED")
    os.environ["CHB_FUSED"] = "0"
    try:
        b = _lib.Context(0)
    finally:
        if old is None:
            del os.environ["CHB_FUSED"]
        else:
            os.environ["CHB_FUSED"] = old
    try:
        b.set_samples(X)
        got_l, its_l, ch_l, mind_l = b.fit_cluster(B, initial, perms, m, 3, batch=batch, want_min_dist=True)
        assert b.counter("fused_enabled") == 0
    finally:
        b.close()
    assert its_l == its and np.array_equal(ch_l, ch) and np.array_equal(got_l, got)
    assert np.allclose(mind_l, mind, rtol=0, atol=QP_TOL, equal_nan=True)


def test_fused_kernel_64bit_row_pointers(O):
    """The m <= 5 fused kernel addresses the rows of X by 32-bit byte offsets below 4 GiB; the 64-bit-pointer
    instantiation (what a larger sample matrix gets) is selected here by CHB_FUSED_PTR64=1 in a child process (the
    switch is read once per process) and must give the same fit."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r)\n"
        "import chbin_amd\n"
        "from chbin_amd import _lib, synth\n"
        "from oracle import oracle as O\n"
        "X, initial, _ = synth.make_synthetic(1200, 136, 6, S=1, seed=5, sigma=6e-3, mix=0.5, n_seed=8)\n"
        "perms = synth.draw_permutations(initial, 3, seed=0)\n"
        "want, its_o, ch_o = O.fit_cluster(X, 6, initial, perms, 5, 3)\n"
        "c = _lib.Context(0); c.set_samples(X)\n"
        "got, its, ch = c.fit_cluster(6, initial, perms, 5, 3, batch=300)\n"
        "assert c.counter('fused_enabled') == 1\n"
        "assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)\n"
        "print('ok')\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, CHB_FUSED_PTR64="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


def test_fit_cluster_rejects_bad_permutations(ctx, O):
    """C ABI: an out-of-range entry in ANY sweep's permutation is reported before anything runs; a sweep that lists a
    sample twice is rejected; the context stays usable and returns the oracle's labels afterwards."""
    from chbin_amd import _lib
    X, initial, _ = _synth(600, 40, 4, seed=2, sigma=8e-3, mix=0.4, n_seed=6)
    perms = _perms(initial, 3)
    ctx.set_samples(X)
    bad = perms.copy()
    bad[2, 7] = len(X)                       # last sweep: would only be reached after two sweeps have run
    with pytest.raises(_lib.ChbError, match="out of range"):
        ctx.fit_cluster(4, initial, bad, 5, 3)
    bad = perms.copy()
    bad[0, 5] = -1
    with pytest.raises(_lib.ChbError, match="out of range"):
        ctx.fit_cluster(4, initial, bad, 5, 3)
    dup = perms.copy()
    dup[0, 11] = dup[0, 3]
    with pytest.raises(_lib.ChbError, match="twice"):
        ctx.fit_cluster(4, initial, dup, 5, 3)
    want, its_o, _ = O.fit_cluster(X, 4, initial, perms, 5, 3)
    got, its, _ = ctx.fit_cluster(4, initial, perms, 5, 3)
    assert its == its_o and np.array_equal(got, want)


def _ctx_env(env):
    from chbin_amd import _lib
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


@pytest.mark.parametrize("m", [5, 15])
def test_segmented_giant_bin(O, m):
    """One bin far larger than the rest (30k of 40k contigs: 938 member tiles, the others ~50): the shortlist stage cuts it
    into segments (two extra launches: per-segment best lists, then shortlists against their union's m-th best).
    Whole fits must equal a context that never segments (CHB_SEGMENTS=0); the selection under the final labels must equal
    the brute-force kernel's lists; the segment launches must really have run."""
    from chbin_amd import _lib
    N, D, Bt = 40_000, 136, 24
    X, initial_t, true_t = _synth(N, D, Bt, seed=5, sigma=2e-3, mix=0.2)
    remap = np.array([0] * 18 + [1, 2, 3, 4, 5, 6])
    B = 7
    initial = np.where(initial_t >= 0, remap[np.maximum(initial_t, 0)], -1).astype(np.int64)
    true = remap[true_t]
    perms = _perms(initial, 2)
    a = _ctx_env({"CHB_TILE_SKIP": "0"})   # (the persistent base pack then serves every batch: segments on its regions)
    try:
        a.set_samples(X)
        got, its, changed = a.fit_cluster(B, initial, perms, m, 2)
        assert a.counter("pack_incremental_batches") == a.fit_stats()["batches"]
        seg_batches = a.counter("segment_batches")
        overflow = a.counter("prefilter_overflow")
        pool_state, pool_batches = a.counter("pool_state"), a.counter("pool_batches")
        assert a.counter("shortlist_short") == 0   # the product build's check of the shortlist stage's contract
        rng = np.random.default_rng(2)
        q = rng.choice(np.flatnonzero(initial < 0), 300, replace=False)
        lists = a.topm_per_bin(got, B, m, q)
        seg_topm = a.counter("segment_batches")
    finally:
        a.close()
    assert seg_batches > 0, "the giant bin was never segmented"
    if m <= 8:
        assert pool_state == -1 and 0 < pool_batches < 8, (pool_state, pool_batches)   # (pools tried, found wanting, dropped)
    else:
        assert pool_batches == 0                                                       # (the 16-entry builds take no pools)
    assert (got == true).mean() > 0.95
    b = _ctx_env({"CHB_SEGMENTS": "0"})
    try:
        b.set_samples(X)
        want, its_w, changed_w = b.fit_cluster(B, initial, perms, m, 2)
        assert b.counter("segment_batches") == 0
    finally:
        b.close()
    assert its == its_w and np.array_equal(changed, changed_w) and np.array_equal(got, want)
    # (m = 15 overflows ~0.4 % of its pairs on any data.  Round 5: the giant bin is 18 clusters under one label, so most
    #  contigs' nearest bin centre says nothing about where they lie and the threshold pools give them loose thresholds;
    #  the fit notices -- candidates per pair of a batch above m + 3 -- and goes back to the two-sweep launch, one batch
    #  late under the look-ahead: the long shortlists of those two or three early batches are what the bound allows for)
    assert overflow <= (1e-3 if m <= 5 else 4e-2) * its * perms.shape[1] * B
    c = _brute_ctx()
    try:
        c.set_samples(X)
        want_lists = c.topm_per_bin(got, B, m, q)
    finally:
        c.close()
    for g, w_ in zip(lists, want_lists):
        assert np.array_equal(g, w_)
    # the oracle replays the LAST contigs of sweep 2 (the giant bin is at full size there)
    tail = perms[its - 1][-12:]
    first = None
    d = _lib.Context(0)
    try:
        d.set_samples(X)
        first, _, _ = d.fit_cluster(B, initial, perms[:1], m, 1)
    finally:
        d.close()
    for k, j in enumerate(tail):
        lab_now = got.copy()
        lab_now[tail[k:]] = first[tail[k:]] if its == 2 else initial[tail[k:]]
        lab_j, _ = O.sweep(X, B, lab_now, np.array([j]), m)
        assert lab_j[j] == got[j]


@pytest.mark.parametrize("D,S,m", [(140, 5, 5), (146, 10, 5), (140, 5, 8), (141, 5, 15)])
def test_tile_skipping_exact(O, D, S, m):
    """Several coverage columns: the base shortlist launch orders a bin's members in norm shells, seats the queries by
    their nearest bin centre and ends a (bin, sweep) run of tiles as soon as no query of the workgroup can find a member of
    its top m in the rest.  Whole fits must equal a context that never skips (CHB_TILE_SKIP=0); the selection under the
    final labels must equal the brute-force kernel's lists; the oracle replays contigs of the last sweep; and tiles must
    really have been skipped."""
    from chbin_amd import _lib
    N, B = 48_000, 24
    X, initial, true = _synth(N, D, B, S=S, seed=11, sigma=2e-3, mix=0.2)
    perms = _perms(initial, 3)
    a = _lib.Context(0)
    try:
        a.set_samples(X)
        got, its, changed = a.fit_cluster(B, initial, perms, m, 3)
        state, unloaded = a.counter("tile_skip_state"), a.counter("tile_unloaded")
        skipped, seen = a.counter("tile_skipped"), a.counter("tile_seen")
        overflow = a.counter("prefilter_overflow")
        assert a.counter("shortlist_short") == 0   # the product build's check of the shortlist stage's contract
        rng = np.random.default_rng(4)
        q = rng.choice(np.flatnonzero(initial < 0), 300, replace=False)
        lists = a.topm_per_bin(got, B, m, q)
    finally:
        a.close()
    # (ten coverage columns leave this generator's bins too round to skip much, and the 15th neighbour is too far to
    #  bound anything away: the fit may turn the skipping off)
    if D < 146 and m <= 8:
        assert state == 1 and unloaded > 0.1 * (seen + unloaded), (state, unloaded, skipped, seen)
    else:
        assert state in (1, -1) and seen > 0
    b = _ctx_env({"CHB_TILE_SKIP": "0"})
    try:
        b.set_samples(X)
        want, its_w, changed_w = b.fit_cluster(B, initial, perms, m, 3)
        assert b.counter("tile_unloaded") == 0 and b.counter("tile_seen") == 0
    finally:
        b.close()
    assert its == its_w and np.array_equal(changed, changed_w) and np.array_equal(got, want)
    assert overflow <= (1e-3 if m <= 5 else 2e-2) * its * perms.shape[1] * B
    c = _brute_ctx()
    try:
        c.set_samples(X)
        want_lists = c.topm_per_bin(got, B, m, q)
    finally:
        c.close()
    for g, w_ in zip(lists, want_lists):
        assert np.array_equal(g, w_)
    # the oracle on the converged labels: a fixed point of the reference sweep (frozen evaluation of a sample)
    if changed[its - 1] == 0:
        sample = rng.choice(np.flatnonzero(initial < 0), 96, replace=False)
        bb, _ = O.eval_frozen_mt(X, B, got, sample, m, 8)
        assert np.array_equal(bb, got[sample])
    # and it replays the LAST contigs of the last sweep run
    tail = perms[its - 1][-8:]
    prev = initial
    if its > 1:
        d = _lib.Context(0)
        try:
            d.set_samples(X)
            prev, _, _ = d.fit_cluster(B, initial, perms[:its - 1], m, its - 1)
        finally:
            d.close()
    for k, j in enumerate(tail):
        lab_now = got.copy()
        lab_now[tail[k:]] = prev[tail[k:]]
        lab_j, _ = O.sweep(X, B, lab_now, np.array([j]), m)
        assert lab_j[j] == got[j]


def test_short_shortlist_is_an_error_not_a_wrong_hull(tmp_path):
    """Every fused hull launch of the PRODUCT build checks the shortlist stage's contract (each (position, bin) base
    shortlist holds at least min(m, bin size) candidates, all of them sample indices) and the fit fails loudly at the
    sweep's end if a pair broke it -- in round 3 a short shortlist was a GPU memory fault or a silently wrong hull.  The
    developer library can hand the hull kernels one truncated shortlist (CHB_SL_INJECT_SHORT=<n-th batch>): the same
    kernels must turn it into CHB_ESTATE.  Runs in a child process (its own library); skipped where the developer
    library has not been built (make -C ch-bin_amd/csrc DEV=1)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = os.path.join(root, "ch-bin_amd", "libchbin_hip_dev.so")
    if not os.path.exists(dev):
        pytest.skip("developer library not built")
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import chbin_amd
from chbin_amd import _lib, synth
for m in (5, 15):
    X, initial, _ = synth.make_synthetic(6000, 136, 12, seed=3)
    perms = synth.draw_permutations(initial, 2, seed=0)
    ctx = _lib.Context(0)
    ctx.set_samples(X)
    try:
        ctx.fit_cluster(12, initial, perms, m, 2, batch=1024)
    except _lib.ChbError as e:
        assert "shortlists of this sweep came out short" in str(e), e
        assert ctx.counter("shortlist_short") >= 1
        print("caught", m)
    else:
        raise SystemExit("the truncated shortlist went unnoticed (m = %%d)" %% m)
    # the next fit on the same context is clean again
    lab, its, _ = ctx.fit_cluster(12, initial, perms, m, 2, batch=1024)
    assert ctx.counter("shortlist_short") == 0
    ctx.close()
""" % root
    script = tmp_path / "inject.py"
    script.write_text(code)
    env = dict(os.environ, CHBIN_LIB=dev, CHB_SL_INJECT_SHORT="2")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
    assert p.stdout.count("caught") == 2


@pytest.mark.parametrize("N,D,B,m,iters,sigma,mix,n_seed,batch", [
    (6000, 136, 12, 5, 4, 6e-3, 0.5, 12, 512),     # overlapping bins: labels keep changing (appends, holes, moved regions)
    (9000, 136, 16, 5, 3, 1.5e-3, 0.0, None, 1024),
    (2500, 140, 9, 15, 3, 4e-3, 0.3, 20, 400),     # 16-lane fused kernel, five coverage columns
])
def test_persistent_pack_equals_rebuild(O, N, D, B, m, iters, sigma, mix, n_seed, batch):
    """The member pack of the shortlist stage kept across the batches of a fit (a batch's members become holes, a commit puts
    them back in place or appends them to their new bin, full regions move: algorithm.py:50,60 as an update) against the
    rebuild of CSR and pack at every batch start (CHB_PACK_INCR=0), and both against the oracle.  CHB_TILE_SKIP=0 keeps the
    tile-skipping builds (which need the rebuild's shell order) out of the way, so the persistent pack serves every batch
    from the first one on."""
    X, initial, _ = _synth(N, D, B, S=5 if D == 140 else 1, seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = _perms(initial, iters)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters)
    a = _ctx_env({"CHB_TILE_SKIP": "0"})
    try:
        a.set_samples(X)
        got, its, ch, mind = a.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
        st = a.fit_stats()
        assert a.counter("pack_incremental_batches") == st["batches"] and a.counter("pack_builds") >= 1
        assert a.counter("shortlist_short") == 0
        got_t, its_t, ch_t = a.fit_cluster(B, initial, perms, m, iters, batch=batch)     # the look-ahead path
        assert a.counter("pack_incremental_batches") > 0
    finally:
        a.close()
    b = _ctx_env({"CHB_TILE_SKIP": "0", "CHB_PACK_INCR": "0"})
    try:
        b.set_samples(X)
        ref, its_r, ch_r, mind_r = b.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
        assert b.counter("pack_incremental_batches") == 0
    finally:
        b.close()
    assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    assert its_t == its_o and np.array_equal(ch_t, ch_o) and np.array_equal(got_t, want)
    assert its_r == its_o and np.array_equal(ref, want)
    mv = initial < 0
    assert np.allclose(mind[mv], mind_r[mv], rtol=0, atol=QP_TOL)
    if mix >= 0.5:
        assert ch_o[1] > 0    # the case really moves contigs after the first sweep


@pytest.mark.parametrize("N,D,B,m,iters,sigma,mix,n_seed,batch,expect", [
    (24000, 136, 24, 5, 3, 1.5e-3, 0.0, None, 2048, "kept"),       # the benchmark generator: pools serve every batch
    (24000, 136, 24, 5, 4, 4.5e-3, 0.3, None, 2048, "any"),        # overlapping bins: labels change in later sweeps (holes, departures)
    (20000, 146, 20, 5, 3, 2e-3, 0.2, None, 4096, "any"),          # ten coverage columns: the tile-skipping pool build
    (5000, 136, 8, 8, 3, 2e-3, 0.2, None, 512, "any"),             # m = 8 builds
    (2400, 140, 8, 15, 3, 2e-3, 0.2, None, 512, "any"),            # m = 15 builds (16-entry lists), five coverage columns
    (6000, 136, 8, 5, 4, 6e-3, 0.6, 12, 512, "any"),               # heavy overlap: long shortlists, the fit drops the pools
])
def test_threshold_pools_equal_two_sweeps(O, N, D, B, m, iters, sigma, mix, n_seed, batch, expect):
    """Round 5: the base shortlist launch takes tau(j, c) from a POOL -- the 32 members of c nearest to the centre of j's home
    bin, one tile per (bin, home bin), kept up by every commit -- and streams the bin once, instead of learning tau in a
    sweep of its own (prefilter_kernels.hip, "threshold pools").  Any m base members bound the m-th nearest distance from
    above, so the selection of distance_matrix.py:47-62 is unchanged: whole fits with the pools must equal a context
    without them (CHB_POOL_TAU=0) and the oracle -- labels, sweep counts, change counts, winning distances -- over several
    sweeps (holes for the open batch's members, departures, arrivals), with and without the look-ahead."""
    X, initial, _ = _synth(N, D, B, S=1 if D <= 136 else (5 if D == 140 else 10), seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = _perms(initial, iters)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, m, iters) if N <= 12000 else (None, None, None)
    from chbin_amd import _lib
    a = _lib.Context(0)
    try:
        a.set_samples(X)
        got, its, ch, mind = a.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
        pool_batches, pool_state = a.counter("pool_batches"), a.counter("pool_state")
        assert a.counter("shortlist_short") == 0
        got_t, its_t, ch_t = a.fit_cluster(B, initial, perms, m, iters, batch=batch)     # the look-ahead path
    finally:
        a.close()
    assert (pool_batches > 0) == (m <= 8)                              # the pools really served batches (m <= 8 builds only)
    if expect == "kept":
        assert pool_state == 1 and pool_batches >= 0.8 * its * (len(perms[0]) / batch)
    b = _ctx_env({"CHB_POOL_TAU": "0"})
    try:
        b.set_samples(X)
        ref, its_r, ch_r, mind_r = b.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
        assert b.counter("pool_batches") == 0
    finally:
        b.close()
    assert its == its_r and np.array_equal(ch, ch_r) and np.array_equal(got, ref)
    assert its_t == its_r and np.array_equal(ch_t, ch_r) and np.array_equal(got_t, ref)
    mv = initial < 0
    assert np.allclose(mind[mv], mind_r[mv], rtol=0, atol=QP_TOL)
    if want is not None:
        assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
    else:
        # the oracle on the converged labels / on a sample of the last sweep's visits (frozen evaluation)
        rng = np.random.default_rng(3)
        if ch[its - 1] == 0:
            sample = rng.choice(np.flatnonzero(initial < 0), 128, replace=False)
            bb, _ = O.eval_frozen_mt(X, B, got, sample, m, 8)
            assert np.array_equal(bb, got[sample])


def test_persistent_pack_rebuilds_under_pressure(tmp_path):
    """The persistent base pack is rebuilt from the labels (holes squeezed out, regions re-sized) when much of its row arena
    has been handed out -- a path ordinary data never reaches (the arena holds 12 N + 1024 B rows).  The developer library
    takes the fill mark from CHB_PACK_REBUILD_AT: with a mark below the initial layout EVERY batch start outside a look-ahead
    window rebuilds.  Labels must equal the oracle's, with the look-ahead on and off.  Child process (its own library);
    skipped where the developer library has not been built."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dev = os.path.join(root, "ch-bin_amd", "libchbin_hip_dev.so")
    if not os.path.exists(dev):
        pytest.skip("developer library not built")
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import chbin_amd
from chbin_amd import _lib, synth
from oracle import oracle as O
X, initial, _ = synth.make_synthetic(5000, 136, 10, seed=21, sigma=6e-3, mix=0.5, n_seed=10)
perms = synth.draw_permutations(initial, 4, seed=0)
want, its_o, ch_o = O.fit_cluster(X, 10, initial, perms, 5, 4)
ctx = _lib.Context(0)
ctx.set_samples(X)
got, its, ch, mind = ctx.fit_cluster(10, initial, perms, 5, 4, batch=400, want_min_dist=True)
builds_a = ctx.counter("pack_builds")
assert its == its_o and np.array_equal(ch, ch_o) and np.array_equal(got, want)
got2, its2, ch2 = ctx.fit_cluster(10, initial, perms, 5, 4, batch=400)
builds_b = ctx.counter("pack_builds")
assert its2 == its_o and np.array_equal(got2, want)
assert ctx.counter("shortlist_short") == 0
print("builds", builds_a, builds_b, ctx.counter("pack_incremental_batches"), ctx.counter("lookahead_batches"))
assert builds_a > 10 and builds_b > 10
ctx.close()
""" % root
    script = tmp_path / "rebuild.py"
    script.write_text(code)
    env = dict(os.environ, CHBIN_LIB=dev, CHB_TILE_SKIP="0", CHB_PACK_REBUILD_AT="1")
    p = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-1500:]
