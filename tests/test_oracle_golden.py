"""The CPU oracle against the fixtures captured from the reference's own Python
(tests/golden/make_golden.py) and against the independent exhaustive enumerator."""
import os

import numpy as np
import pytest

from oracle import oracle as O


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_cdist_bit_exact(golden_dir):
    g = _load(golden_dir, "cdist.npz")
    M = O.cdist(g["X"])
    assert np.array_equal(M, g["M"])  # scipy cdist == sequential non-FMA sum, then sqrt
    assert M[3, 5] == 0.0


def test_find_nearest_matches_reference_selection(golden_dir):
    g = _load(golden_dir, "find_nearest.npz")
    X, labels, m = g["X"], g["labels"], int(g["m"])
    for i, c, want in zip(g["rows"], g["bins"], g["selected_sorted"]):
        cur = labels.copy()
        cur[i] = -1
        got = O.find_nearest_from_cluster(int(c), cur, O.cdist_row(X, int(i)), m)
        want = want[want >= 0]
        assert np.array_equal(np.sort(got), want), (i, c)
        # documented ordering: (distance, index) ascending
        d = O.cdist_row(X, int(i))[got]
        assert np.all(np.diff(d) >= 0)


def test_qp_argument_construction(golden_dir):
    """solve_qp.py:44-51: G' = nearestPD(2XX^T), a = 2Xx, C = [-1 | I], b = [-1, -0...], meq=1."""
    g = _load(golden_dir, "qp_args.npz")
    for k in range(len(g["m"])):
        m = int(g["m"][k])
        P, x = g["P"][k][:m], g["x"][k]
        G_ref, a_ref = g["G"][k][:m, :m], g["a"][k][:m]
        Pm = 2.0 * P @ P.T
        G = O.nearest_positive_definite(Pm)
        scale = np.abs(Pm).max()
        assert np.allclose(G, G_ref, rtol=0, atol=1e-12 * scale)
        assert np.allclose(2.0 * P @ x, a_ref, rtol=1e-14, atol=0)
        C_ref, b_ref = g["C"][k][:m, : m + 1], g["b"][k][: m + 1]
        assert np.array_equal(C_ref, np.hstack([-np.ones((m, 1)), np.eye(m)]))
        assert np.array_equal(b_ref, np.concatenate([[-1.0], np.zeros(m)]))
        assert int(g["meq"][k]) == 1
        # the recorded tuple solved by the restated Goldfarb-Idnani agrees with the enumerator
        alpha = O.gi_solve(G_ref, a_ref, C_ref, b_ref, 1)
        d_gi = np.linalg.norm(alpha @ P - x)
        assert abs(d_gi - O.enum_hull_distance(x, P)) < 1e-7
        assert abs(O.convex_hull_distance(x, P) - g["dist_with_oracle_gi"][k]) < 1e-9


def test_fit_cluster_control_flow(golden_dir):
    """Labels from the reference's own fit_cluster loop == the oracle's restated loop."""
    g = _load(golden_dir, "fit_cluster_flow.npz")
    labels, its, changed = O.fit_cluster(g["X"], int(g["B"]), g["initial"], g["perms"],
                                         int(g["m"]), int(g["max_iter"]))
    assert np.array_equal(labels, g["labels"])
    assert its >= 3 and changed[-1] == 0
    # with the precomputed matrix (InMemDistMatrix=yes semantics) nothing changes
    labels2, _, _ = O.fit_cluster(g["X"], int(g["B"]), g["initial"], g["perms"], int(g["m"]),
                                  int(g["max_iter"]), dm=O.cdist(g["X"]))
    assert np.array_equal(labels2, labels)


def test_fit_cluster_flow_does_not_depend_on_the_solver(golden_dir):
    """fit_cluster_flow.npz came from the reference's loop with a stand-in quadprog that answers with the ORACLE's own
    Goldfarb-Idnani -- circular for the solver stage.  fit_cluster_flow_slsqp.npz is the same reference loop on the same
    inputs with scipy's SLSQP answering instead (tests/golden/make_golden_second_solver.py, 22,680 QP calls, all
    converged): the labels are identical, so what the fixture pins is not an artefact of the oracle's solver."""
    g = _load(golden_dir, "fit_cluster_flow.npz")
    s = _load(golden_dir, "fit_cluster_flow_slsqp.npz")
    assert int(s["not_converged"]) == 0 and int(s["qp_calls"]) > 20000
    assert float(s["max_eq_violation"]) < 1e-12 and float(s["min_alpha"]) > -1e-12
    assert np.array_equal(s["labels_slsqp"], g["labels"])
    labels, _, _ = O.fit_cluster(g["X"], int(g["B"]), g["initial"], g["perms"], int(g["m"]), int(g["max_iter"]))
    assert np.array_equal(labels, s["labels_slsqp"])


def test_distances_against_numbers_a_second_solver_produced(golden_dir):
    """Distances that a solver OTHER than the oracle's Goldfarb-Idnani produced, through the reference's own glue
    (hull_distance.py:7-35 imported unchanged, scipy's SLSQP answering quadprog.solve_qp on the reference's tuple;
    tests/golden/make_golden_second_solver.py): the 24 captured hull problems of qp_args.npz and every 45th of the 22,680
    (contig, bin) evaluations of the reference's loop on the 600-contig case.  North-star tolerance 1e-5; observed
    maxima 2.2e-9 (qp_args: random points in D = 136, SLSQP's own stopping accuracy) and 4e-16 (loop problems)."""
    g = _load(golden_dir, "qp_args.npz")
    s = _load(golden_dir, "qp_args_slsqp.npz")
    worst = 0.0
    for k in range(len(g["m"])):
        m = int(g["m"][k])
        d = O.convex_hull_distance(g["x"][k], g["P"][k][:m])
        worst = max(worst, abs(d - s["dist_with_slsqp"][k]))
        assert abs(O.enum_hull_distance(g["x"][k], g["P"][k][:m]) - s["dist_with_slsqp"][k]) < 1e-5
    assert worst < 1e-5 and worst < 1e-8, worst
    f = _load(golden_dir, "fit_cluster_flow.npz")
    t = _load(golden_dir, "fit_cluster_flow_slsqp.npz")
    assert len(t["loop_query"]) == 504
    X, worst = f["X"], 0.0
    for q, h, c, d in zip(t["loop_query"], t["loop_hull"], t["loop_bin"], t["loop_dist_slsqp"]):
        h = h[h >= 0]
        worst = max(worst, abs(O.convex_hull_distance(X[q], X[h]) - d))
    assert worst < 1e-5 and worst < 1e-12, worst


@pytest.mark.parametrize("seed", range(4))
def test_gi_vs_enumerator_random(seed):
    rng = np.random.default_rng(seed)
    for trial in range(400):
        m = int(rng.integers(1, 9))
        D = int(rng.integers(2, 60))
        P = rng.random((m, D))
        kind = trial % 5
        if kind == 0:
            x = rng.random(D)
        elif kind == 1:
            x = rng.dirichlet(np.ones(m)) @ P
        elif kind == 2:
            x = rng.random(D) * 3 - 1
        elif kind == 3:
            x = P[rng.integers(m)] + 1e-3 * rng.standard_normal(D)
        else:
            if m > 1:
                P[m - 1] = P[0]
            x = rng.random(D)
        d_gi = O.convex_hull_distance(x, P)
        d_en = O.enum_hull_distance(x, P)
        assert abs(d_gi - d_en) < 1e-8 * max(1.0, d_en), (seed, trial, m, D, kind)


def test_closed_forms():
    rng = np.random.default_rng(5)
    D = 136
    x = rng.random(D) / D
    p = rng.random((1, D)) / D
    # m = 1: Euclidean distance
    assert abs(O.convex_hull_distance(x, p) - np.linalg.norm(x - p[0])) < 1e-15
    # m = 2: clamped segment projection
    P = rng.random((2, D)) / D
    t = np.clip(np.dot(x - P[0], P[1] - P[0]) / np.dot(P[1] - P[0], P[1] - P[0]), 0, 1)
    want = np.linalg.norm(P[0] + t * (P[1] - P[0]) - x)
    assert abs(O.convex_hull_distance(x, P) - want) < 1e-12
    # empty hull guard
    d, st = O.convex_hull_distance(x, np.zeros((0, D)), return_status=True)
    assert np.isinf(d) and st == 2
    # invariance: vertex permutation and translation
    P = rng.random((5, D)) / D
    d0 = O.convex_hull_distance(x, P)
    assert abs(O.convex_hull_distance(x, P[::-1].copy()) - d0) < 1e-12
    sh = rng.random(D)
    assert abs(O.convex_hull_distance(x + sh, P + sh) - d0) < 1e-10


def test_affine_hull_distance_matches_reference_formula():
    """hull_distance.py:69-87 restated literally with numpy/scipy vs the oracle's C version."""
    import scipy.linalg

    def ref(q, P):
        mean = P.mean(axis=0)
        basis = scipy.linalg.orth((P - mean).T)
        if basis.shape[1] == 0:
            return np.linalg.norm(q - mean)
        proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
        return np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (q - mean))

    rng = np.random.default_rng(3)
    for t in range(300):
        m = int(rng.integers(1, 12))
        D = int(rng.integers(2, 60))
        P = rng.random((m, D))
        x = rng.random(D)
        if t % 3 == 0 and m > 2:
            P[2] = 0.5 * (P[0] + P[1])
        assert abs(O.affine_hull_distance(x, P) - ref(x, P)) < 1e-11
        assert O.affine_hull_distance(x, P) <= O.convex_hull_distance(x, P) + 1e-9


def test_affine_hull_distance_16_vertices_fixture(golden_dir):
    """tests/golden/affine_16_vertices.npz (made by tests/golden/make_affine_case.py): 16 vertices in D = 40 whose
    centred matrix has one pure-noise singular value.  scipy.linalg.orth drops it (rank 15); a Gram-Schmidt
    restatement with the cutoff on residual norms kept it and returned 0.024124 instead of 0.024251 (found by
    tools/fuzz_fit.py in round 2, where every GPU path agreed with the scipy value)."""
    import scipy.linalg
    g = np.load(os.path.join(golden_dir, "affine_16_vertices.npz"))
    x, P = g["x"], g["P"]
    mean = P.mean(axis=0)
    basis = scipy.linalg.orth((P - mean).T)
    assert basis.shape[1] == int(g["rank"]) == 15
    proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
    live = np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (x - mean))
    assert abs(live - float(g["expected"])) < 1e-13
    assert abs(O.affine_hull_distance(x, P) - float(g["expected"])) < 1e-12


def test_affine_hull_distance_many_vertices_with_common_offset():
    """The shape that separates a faithful restatement of scipy.linalg.orth from Gram-Schmidt with a cutoff on
    residual norms: 12 .. 16 vertices whose features share a large offset (coverage columns of 0.2 next to k-mer
    frequencies of 0.004).  The centred rows sum to zero only up to rounding; orth drops that direction (its
    singular value is far below sigma_max * eps * max(m, D)), a residual-norm cutoff can keep it and return a
    distance that is too small (found by tools/fuzz_fit.py in round 2)."""
    import scipy.linalg

    def ref(q, P):
        mean = P.mean(axis=0)
        basis = scipy.linalg.orth((P - mean).T)
        proj = basis @ np.linalg.inv(basis.T @ basis) @ basis.T
        return np.linalg.norm((np.eye(proj.shape[0]) - proj) @ (q - mean))

    rng = np.random.default_rng(11)
    for t in range(200):
        m = int(rng.integers(12, 17))
        # (D >= 40: with fewer features the cutoff sigma_max * eps * max(m, D) sits below the rounding noise itself and
        #  the reference formula has no stable value -- LAPACK's noise direction is as good as anybody's)
        D = int(rng.choice([40, 100, 136]))
        base = np.concatenate([np.full(D // 8, 0.2), np.full(D - D // 8, 0.004)])
        P = base + 0.004 * rng.standard_normal((m, D))
        x = base + 0.01 * rng.standard_normal(D)
        want = ref(x, P)
        got = O.affine_hull_distance(x, P)
        assert abs(got - want) <= 1e-12 + 1e-10 * want, (t, m, D, got, want)
