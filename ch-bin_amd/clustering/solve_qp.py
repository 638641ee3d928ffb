"""Mirror of ch_bin/core/clustering/solve_qp.py.

In the reference `solve_qp` (solve_qp.py:96-132) hands a dense (P, q, G, h, A, b) QP to quadprog or cvxopt.
Its only callers are the two hull distances of hull_distance.py, which build exactly two forms:

  * convex hull  (hull_distance.py:17-33):  P = 2 X X^T, q = -2 X x, G = -I, h = 0, A = 1^T, b = 1
  * affine hull  (hull_distance.py:48-64):  the same P, q, A, b and an inequality block of ZERO rows
                                            (G = np.zeros((0, m)), h = np.zeros(0); None / None is accepted too)

On the HIP path these QPs are normally never materialised (the kernel builds the Gram of a hull and solves it
in one pass).  This function keeps the seam callable: it recognises the two forms and solves them on the GPU
with the same kernel.  The kernel wants points, not (P, q); any point set with Gram P / 2 and inner products
-q / 2 with the query poses the same problem, so one is rebuilt from the symmetric eigen-decomposition of P
(m x m, host side -- the counterpart of the reference's nearest-PD preprocessing, solve_qp.py:44) and the QP
itself runs in `chb_hull_distance_points`.  Any other QP raises NotImplementedError: nothing in the reference
poses one.  'quadprog', 'cvxopt' and 'hip' all select the HIP solver (solve_qp.py:125-132 name contract).
"""
import numpy as np

SOLVERS = ("quadprog", "cvxopt", "hip")


def check_solver(solver: str) -> None:
    if solver not in SOLVERS:
        raise NotImplementedError(f"Unknown solver {solver}")  # solve_qp.py:132


def _recognise(mat_p, vec_q, mat_g, vec_h, mat_a, vec_b):
    """'convex' / 'affine' for the two forms hull_distance.py builds, else None."""
    m = mat_p.shape[0]
    if mat_p.shape != (m, m) or vec_q.shape != (m,):
        return None
    if mat_a is None or vec_b is None:
        return None
    a = np.asarray(mat_a, dtype=np.float64).reshape(-1, m) if np.size(mat_a) % m == 0 else None
    if a is None or a.shape[0] != 1 or not np.all(a == 1.0):
        return None
    if np.asarray(vec_b, dtype=np.float64).reshape(-1).tolist() != [1.0]:
        return None
    # "no inequality" is G = h = None or, as hull_distance.py:54-55 really passes it, a block of zero rows
    # (np.zeros((0, n)), np.zeros(0))
    no_g = mat_g is None or np.size(mat_g) == 0
    no_h = vec_h is None or np.size(vec_h) == 0
    if no_g and no_h:
        return "affine"
    if no_g or no_h:
        return None
    g = np.asarray(mat_g, dtype=np.float64)
    hv = np.asarray(vec_h, dtype=np.float64).reshape(-1)
    if g.shape == (m, m) and np.array_equal(g, -np.eye(m)) and hv.shape == (m,) and not np.any(hv):
        return "convex"
    return None


def _surrogate_points(mat_p, vec_q):
    """Rows Z (m x m) and a query z with Z Z^T = P / 2 and Z z = -q / 2 (least squares if q is not in range)."""
    gram = 0.25 * (mat_p + mat_p.T)                       # P / 2, symmetrised like solve_qp.py:79
    w, v = np.linalg.eigh(gram)
    z_rows = v * np.sqrt(np.clip(w, 0.0, None))           # Z = V diag(sqrt(w)):  Z Z^T = gram
    z, *_ = np.linalg.lstsq(z_rows, -0.5 * vec_q, rcond=None)
    return z_rows, z


def solve_qp(mat_p, vec_q, mat_g=None, vec_h=None, mat_a=None, vec_b=None, solver: str = "quadprog"):
    """solve_qp.py:96-132 for the two QP forms the reference itself poses; returns the weight vector alpha."""
    check_solver(solver)
    mat_p = np.asarray(mat_p, dtype=np.float64)
    vec_q = np.asarray(vec_q, dtype=np.float64).reshape(-1)
    form = _recognise(mat_p, vec_q, mat_g, vec_h, mat_a, vec_b)
    if form is None:
        raise NotImplementedError(
            "solve_qp on the HIP path covers the two forms hull_distance.py builds (simplex-constrained and "
            "sum-to-one-only least squares); a general QP is not part of the accelerated path")
    from .._lib import default_context
    z_rows, z = _surrogate_points(mat_p, vec_q)
    with default_context().using_metric(form) as ctx:   # (the caller's metric is put back afterwards)
        _, alpha = ctx.hull_distance_points(z, z_rows, want_alpha=True)
    return alpha
