"""Replays the reference's own fit_cluster loop on the 600-contig fixture with a SECOND, independently written stand-in
for quadprog.solve_qp -- scipy's SLSQP on the exact (G, a, C, b, meq) tuple the reference hands over -- and stores the
labels it returns.  fit_cluster_flow.npz was produced with a stand-in that answers with the oracle's own
Goldfarb-Idnani (make_golden.py), so oracle == fixture was circular for the solver stage; this file shows that the
fixture's labels do not depend on which correct solver answered.

Run ONLY in the build container (needs /root/reference):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_second_solver.py
Writes tests/golden/fit_cluster_flow_slsqp.npz (arrays only; nothing of the reference travels).
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

    z = np.load(os.path.join(HERE, "fit_cluster_flow.npz"))
    X, initial = z["X"], z["initial"]
    M = ref_dm.create_in_mem_distance_matrix(X)
    np.random.seed(0)  # ch_bin.py:22
    labels = ref_alg.fit_cluster(X, int(z["B"]), initial, M, num_neighbors=int(z["m"]), max_iterations=int(z["max_iter"]),
                                 metric="convex", qp_solver="quadprog")
    labels = np.asarray(labels, dtype=np.int64)
    same = bool(np.array_equal(labels, z["labels"]))
    np.savez_compressed(os.path.join(HERE, "fit_cluster_flow_slsqp.npz"), labels_slsqp=labels,
                        qp_calls=STATS["calls"], not_converged=STATS["not_converged"],
                        max_eq_violation=STATS["max_eq_violation"], min_alpha=STATS["min_x"])
    print("second-solver replay:", STATS, "labels equal to fit_cluster_flow.npz:", same,
          "differing:", int((labels != z["labels"]).sum()))


if __name__ == "__main__":
    main()
