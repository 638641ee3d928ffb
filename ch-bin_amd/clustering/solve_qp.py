"""Mirror of ch_bin/core/clustering/solve_qp.py.

In the reference `solve_qp` (solve_qp.py:96-132) dispatches a generic (P, q, G, h, A, b) QP to
quadprog or cvxopt.  On the HIP path the QP is never materialised on the host: the kernel
(csrc/qp_kernels.hip) builds the shifted Gram of a hull and solves the simplex-constrained
least-squares problem in one pass, so the only thing this module keeps from the reference is the
solver-name contract of solve_qp.py:125-132: 'quadprog' and 'cvxopt' stay legal values of
AlgoQpSolver (both now mean "the HIP solver", as does the new explicit value 'hip'); anything else
raises NotImplementedError exactly like the reference.
"""
SOLVERS = ("quadprog", "cvxopt", "hip")


def check_solver(solver: str) -> None:
    if solver not in SOLVERS:
        raise NotImplementedError(f"Unknown solver {solver}")  # solve_qp.py:132


def solve_qp(mat_p, vec_q, mat_g, vec_h, mat_a, vec_b, solver: str = "quadprog"):
    """solve_qp.py:96.  Only the unit-simplex form that hull_distance.py:17-33 constructs
    (A = 1^T, b = 1, G = -I, h = 0) exists on the GPU, and there it is fused with the Gram build;
    a free-standing generic QP is not part of the accelerated path."""
    check_solver(solver)
    raise NotImplementedError(
        "generic solve_qp is not exposed by the HIP path: use calculate_distance / "
        "convex_hull_distance (the QP is built and solved inside the kernel)")
