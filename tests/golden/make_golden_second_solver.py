"""Replays the reference's own fit_cluster loop on the 600-contig fixture with a SECOND, independently written stand-in
for quadprog.solve_qp -- scipy's SLSQP on the exact (G, a, C, b, meq) tuple the reference hands over -- and stores the
labels it returns.  fit_cluster_flow.npz was produced with a stand-in that answers with the oracle's own
Goldfarb-Idnani (make_golden.py), so oracle == fixture was circular for the solver stage; this file shows that the
fixture's labels do not depend on which correct solver answered.

Round 5: it also stores NUMBERS a solver other than the oracle's produced, so that no distance check has to lean on the
oracle's own Goldfarb-Idnani: (1) for the 24 hull problems of qp_args.npz the value the reference's own
calculate_distance (hull_distance.py:7-35, imported unchanged) returns with SLSQP answering quadprog.solve_qp
(`dist_with_slsqp`), and (2) for every 45th of the 22,680 (contig, bin) evaluations of the loop replay the sample index,
the selected member indices (distance_matrix.py:47-62, as the reference's loop passed them on) and the distance the
reference's calculate_distance returned (`loop_query`, `loop_hull`, `loop_bin`, `loop_dist_slsqp`: 504 problems on the
rows of fit_cluster_flow.npz's X).

Run ONLY in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_second_solver.py
Writes tests/golden/fit_cluster_flow_slsqp.npz and tests/golden/qp_args_slsqp.npz (arrays only; nothing of the
reference travels).
"""
import os
import sys
import types

import numpy as np
from scipy.optimize import minimize

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
sys.dont_write_bytecode = True

STATS = {"calls": 0, "max_eq_violation": 0.0, "min_x": 0.0, "not_converged": 0}


def slsqp_solve_qp(G, a, C, b, meq):
    """min 1/2 x'Gx - a'x  s.t.  C[:, :meq]'x == b[:meq],  C[:, meq:]'x >= b[meq:]   (quadprog's convention).
    The variables are scaled by nothing: the problems are m <= 5 simplex QPs with O(1e-3) Gram entries, so the
    objective is scaled by 1 / max|G| to give SLSQP's stopping rule something to bite on."""
    G = np.asarray(G, dtype=np.float64); a = np.asarray(a, dtype=np.float64)
    C = np.asarray(C, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    n = len(a)
    sc = 1.0 / max(np.abs(G).max(), 1e-300)
    cons = []
    for k in range(C.shape[1]):
        ck, bk = C[:, k].copy(), float(b[k])
        cons.append({"type": "eq" if k < meq else "ineq", "fun": (lambda x, ck=ck, bk=bk: ck @ x - bk),
                     "jac": (lambda x, ck=ck: ck)})
    # feasible start: the first equality is -sum(x) = -1 for the hull QP (solve_qp.py:46-49)
    x0 = np.full(n, 1.0 / n)
    best = None
    for _ in range(3):
        r = minimize(lambda x: sc * (0.5 * x @ G @ x - a @ x), x0, jac=lambda x: sc * (G @ x - a), method="SLSQP",
                     constraints=cons, options={"ftol": 1e-15, "maxiter": 500})
        best = r
        if r.success:
            break
        x0 = r.x
    STATS["calls"] += 1
    STATS["not_converged"] += 0 if best.success else 1
    x = best.x
    for k in range(meq):
        STATS["max_eq_violation"] = max(STATS["max_eq_violation"], abs(C[:, k] @ x - b[k]))
    STATS["min_x"] = min(STATS["min_x"], float(x.min()))
    return (x,)


def main():
    numba = types.ModuleType("numba")
    numba.njit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = numba
    sys.modules["cvxopt"] = types.ModuleType("cvxopt")
    quadprog = types.ModuleType("quadprog")
    quadprog.solve_qp = slsqp_solve_qp
    sys.modules["quadprog"] = quadprog

    import logging
    logging.disable(logging.CRITICAL)
    from ch_bin.core.clustering import algorithm as ref_alg
    from ch_bin.core.clustering import distance_matrix as ref_dm
    ref_alg.tqdm = lambda it, **kw: it

    from ch_bin.core.clustering import hull_distance as ref_hd

    # ---- (1) the 24 captured hull problems through the reference's own glue, SLSQP answering
    q = np.load(os.path.join(HERE, "qp_args.npz"))
    d_slsqp = np.array([ref_hd.calculate_distance(q["x"][k], q["P"][k][: int(q["m"][k])], "quadprog", "convex")
                        for k in range(len(q["m"]))], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "qp_args_slsqp.npz"), dist_with_slsqp=d_slsqp)
    print("qp_args problems: max |slsqp - oracle GI| =", float(np.abs(d_slsqp - q["dist_with_oracle_gi"]).max()))
    for k in STATS:
        STATS[k] = 0 if k != "max_eq_violation" and k != "min_x" else 0.0

    z = np.load(os.path.join(HERE, "fit_cluster_flow.npz"))
    X, initial = z["X"], z["initial"]
    M = ref_dm.create_in_mem_distance_matrix(X)

    # ---- (2) every 45th (contig, bin) evaluation of the loop: what was selected and what distance came back.  The two
    # names algorithm.py:55-56 calls are wrapped in ITS namespace (recorders only: arguments and results pass through)
    SAMPLE_EVERY = 45
    rec = {"n": 0, "last": None, "query": [], "hull": [], "bin": [], "dist": []}
    m_fix = int(z["m"])
    orig_find, orig_calc = ref_alg.find_nearest_from_cluster, ref_alg.calculate_distance

    def find_rec(c, curr_bins, distance_row, num_neighbors):
        idx = orig_find(c, curr_bins, distance_row, num_neighbors)
        rec["last"] = (int(c), np.array(idx, dtype=np.int64))
        return idx

    def calc_rec(x, pts, qp_solver, metric):
        d = orig_calc(x, pts, qp_solver, metric)
        if rec["n"] % SAMPLE_EVERY == 0:
            c, idx = rec["last"]
            i = int(np.flatnonzero((X == x).all(axis=1))[0])
            assert np.array_equal(X[idx], pts)
            pad = np.full(m_fix, -1, dtype=np.int64); pad[: len(idx)] = idx
            rec["query"].append(i); rec["hull"].append(pad); rec["bin"].append(c); rec["dist"].append(float(d))
        rec["n"] += 1
        return d

    ref_alg.find_nearest_from_cluster, ref_alg.calculate_distance = find_rec, calc_rec
    np.random.seed(0)  # ch_bin.py:22
    labels = ref_alg.fit_cluster(X, int(z["B"]), initial, M, num_neighbors=int(z["m"]), max_iterations=int(z["max_iter"]),
                                 metric="convex", qp_solver="quadprog")
    labels = np.asarray(labels, dtype=np.int64)
    same = bool(np.array_equal(labels, z["labels"]))
    np.savez_compressed(os.path.join(HERE, "fit_cluster_flow_slsqp.npz"), labels_slsqp=labels,
                        qp_calls=STATS["calls"], not_converged=STATS["not_converged"],
                        max_eq_violation=STATS["max_eq_violation"], min_alpha=STATS["min_x"],
                        loop_query=np.array(rec["query"], dtype=np.int64), loop_hull=np.array(rec["hull"], dtype=np.int64),
                        loop_bin=np.array(rec["bin"], dtype=np.int64), loop_dist_slsqp=np.array(rec["dist"], dtype=np.float64))
    print("loop evaluations recorded:", len(rec["dist"]), "of", rec["n"])
    print("second-solver replay:", STATS, "labels equal to fit_cluster_flow.npz:", same,
          "differing:", int((labels != z["labels"]).sum()))


if __name__ == "__main__":
    main()
