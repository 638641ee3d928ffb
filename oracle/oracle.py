"""ctypes front-end of the CPU oracle (oracle/chb_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package never imports this module.

Reference citations live in chb_oracle.c next to each function.  Parity status: "parity
unpinned" at the quadprog boundary (see chb_oracle.c header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libchb_oracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (seconds)."""
    src = os.path.join(_HERE, "chb_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.chbo_cdist_row.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _f64p]
        L.chbo_cdist_row.restype = None
        L.chbo_cdist.argtypes = [_f64p, C.c_int64, C.c_int64, _f64p]
        L.chbo_cdist.restype = None
        L.chbo_find_nearest.argtypes = [C.c_int64, _i64p, _f64p, C.c_int64, C.c_int, _i64p]
        L.chbo_find_nearest.restype = C.c_int
        L.chbo_nearest_pd.argtypes = [_f64p, C.c_int, _f64p]
        L.chbo_nearest_pd.restype = None
        L.chbo_gi_solve.argtypes = [C.c_int, _f64p, _f64p, C.c_int, _f64p, _f64p, C.c_int, _f64p,
                                    C.POINTER(C.c_int)]
        L.chbo_gi_solve.restype = C.c_int
        L.chbo_enum_hull_distance.argtypes = [_f64p, _f64p, C.c_int, C.c_int64, _f64p]
        L.chbo_enum_hull_distance.restype = C.c_double
        L.chbo_convex_hull_distance.argtypes = [_f64p, _f64p, C.c_int, C.c_int64, _f64p,
                                                C.POINTER(C.c_int)]
        L.chbo_convex_hull_distance.restype = C.c_double
        L.chbo_affine_hull_distance.argtypes = [_f64p, _f64p, C.c_int, C.c_int64]
        L.chbo_affine_hull_distance.restype = C.c_double
        L.chbo_sweep_metric.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p, C.c_int64,
                                        C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.chbo_sweep_metric.restype = C.c_int64
        L.chbo_fit_cluster_metric.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p,
                                              C.c_int64, C.c_int, C.c_int, C.c_void_p, _i64p, _i64p,
                                              C.c_int]
        L.chbo_fit_cluster_metric.restype = C.c_int
        L.chbo_sweep.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p, C.c_int64,
                                 C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.chbo_sweep.restype = C.c_int64
        L.chbo_fit_cluster.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p,
                                       C.c_int64, C.c_int, C.c_int, C.c_void_p, _i64p, _i64p]
        L.chbo_fit_cluster.restype = C.c_int
        L.chbo_eval_frozen_mt.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_int64, _i64p, _i64p, C.c_int64,
                                          C.c_int, C.c_int, _i64p, _f64p]
        L.chbo_eval_frozen_mt.restype = C.c_int64
        L.chbo_kmer_dim.argtypes = [C.c_int, C.c_void_p]
        L.chbo_kmer_dim.restype = C.c_int64
        L.chbo_kmer_frequencies.argtypes = [C.c_char_p, _i64p, C.c_int64, C.c_int, _f64p, _i64p]
        L.chbo_kmer_frequencies.restype = C.c_int
        _lib = L
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i64(a):
    return np.ascontiguousarray(a, dtype=np.int64)


def cdist_row(X, i):
    X = _f64(X)
    row = np.empty(X.shape[0])
    lib().chbo_cdist_row(X, X.shape[0], X.shape[1], int(i), row)
    return row


def cdist(X):
    X = _f64(X)
    out = np.empty((X.shape[0], X.shape[0]))
    lib().chbo_cdist(X, X.shape[0], X.shape[1], out)
    return out


def find_nearest_from_cluster(c, curr_bins, distance_row, m):
    """distance_matrix.py:47-62; returns indices ordered by (distance, index)."""
    labels = _i64(curr_bins)
    row = _f64(distance_row)
    out = np.empty(max(int(m), 1), dtype=np.int64)
    cnt = lib().chbo_find_nearest(int(c), labels, row, labels.shape[0], int(m), out)
    return out[:cnt].copy()


def nearest_positive_definite(A):
    A = _f64(A)
    out = np.empty_like(A)
    lib().chbo_nearest_pd(A, A.shape[0], out)
    return out


def gi_solve(G, a, Cmat, b, meq):
    """quadprog.solve_qp(G, a, C, b, meq)[0] restated; raises ValueError like quadprog does."""
    G, a, Cmat, b = _f64(G), _f64(a), _f64(Cmat), _f64(b)
    n, q = Cmat.shape
    x = np.empty(n)
    it = C.c_int(0)
    rc = lib().chbo_gi_solve(n, G, a, q, Cmat, b, int(meq), x, C.byref(it))
    if rc == 1:
        raise ValueError("constraints are inconsistent, no solution")
    if rc == 2:
        raise ValueError("matrix G is not positive definite")
    return x


def enum_hull_distance(x, P, return_alpha=False):
    x, P = _f64(x), _f64(P)
    m = P.shape[0]
    alpha = np.zeros(max(m, 1))
    d = lib().chbo_enum_hull_distance(x, P.reshape(m, x.shape[0]), m, x.shape[0], alpha)
    return (d, alpha[:m]) if return_alpha else d


def convex_hull_distance(x, P, return_alpha=False, return_status=False):
    """hull_distance.py:7-35 with solver='quadprog'."""
    x, P = _f64(x), _f64(P)
    m = P.shape[0]
    alpha = np.zeros(max(m, 1))
    st = C.c_int(0)
    d = lib().chbo_convex_hull_distance(x, P.reshape(m, x.shape[0]), m, x.shape[0], alpha, C.byref(st))
    out = (d,)
    if return_alpha:
        out += (alpha[:m],)
    if return_status:
        out += (st.value,)
    return out if len(out) > 1 else d


METRICS = {"convex": 0, "affine": 1, "affine-qp": 1}


def affine_hull_distance(x, P):
    """hull_distance.py:69-87 (== :38-66 mathematically): distance to the affine hull."""
    x, P = _f64(x), _f64(P)
    m = P.shape[0]
    return lib().chbo_affine_hull_distance(x, P.reshape(m, x.shape[0]), m, x.shape[0])


def sweep(X, B, labels, perm, m, dm=None, want_all=False, metric="convex"):
    """algorithm.py:46-60 over perm; mutates and returns labels, plus winning distances."""
    X = _f64(X)
    labels = _i64(labels).copy()
    perm = _i64(perm)
    N, D = X.shape
    mind = np.empty(len(perm))
    alld = np.empty((len(perm), B)) if want_all else None
    dmp = None if dm is None else _f64(dm)
    lib().chbo_sweep_metric(X, N, D, int(B), labels, perm, len(perm), int(m),
                            None if dmp is None else dmp.ctypes.data, mind.ctypes.data,
                            None if alld is None else alld.ctypes.data, METRICS[metric])
    return (labels, mind, alld) if want_all else (labels, mind)


def fit_cluster(X, B, initial_bins, perms, m, max_iter, dm=None, metric="convex"):
    """algorithm.py:12-76; perms = (max_iter, n_move) int64, pre-drawn like algorithm.py:45."""
    X = _f64(X)
    initial = _i64(initial_bins)
    perms = _i64(perms).reshape(max_iter, -1)
    N, D = X.shape
    out = np.empty(N, dtype=np.int64)
    changed = np.zeros(max_iter, dtype=np.int64)
    dmp = None if dm is None else _f64(dm)
    its = lib().chbo_fit_cluster_metric(X, N, D, int(B), initial, perms, perms.shape[1], int(m),
                                        int(max_iter), None if dmp is None else dmp.ctypes.data, out,
                                        changed, METRICS[metric])
    return out, its, changed[:its]


def kmer_dim(k):
    """Number of canonical k-mers (kmer_count.py:65-107 output width; 136 for k = 4)."""
    return int(lib().chbo_kmer_dim(int(k), None))


def kmer_frequencies(seqs, k):
    """seqs: list of bytes/str (one per contig) -> (freq [n, dim] f64, counts [n, dim] i64)."""
    bs = [s.encode() if isinstance(s, str) else bytes(s) for s in seqs]
    offsets = np.zeros(len(bs) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum([len(b) for b in bs])
    dim = kmer_dim(k)
    freq = np.zeros((len(bs), dim), dtype=np.float64)
    counts = np.zeros((len(bs), dim), dtype=np.int64)
    rc = lib().chbo_kmer_frequencies(b"".join(bs), offsets, len(bs), int(k), freq.reshape(-1) if len(bs) else freq,
                                     counts.reshape(-1) if len(bs) else counts)
    if rc != 0:
        raise ValueError("unsupported k")
    return freq, counts


def eval_frozen_mt(X, B, labels, ids, m, nthreads):
    """Multi-threaded timing helper: strict-'>' argmin bin and distance of each contig in `ids`
    against frozen labels (see chbo_eval_frozen_mt).  Returns (best_bin, best_dist)."""
    X = _f64(X)
    labels = _i64(labels)
    ids = _i64(ids)
    bb = np.zeros(len(ids), dtype=np.int64)
    bd = np.zeros(len(ids), dtype=np.float64)
    lib().chbo_eval_frozen_mt(X, X.shape[0], X.shape[1], int(B), labels, ids, len(ids), int(m), int(nthreads), bb, bd)
    return bb, bd
