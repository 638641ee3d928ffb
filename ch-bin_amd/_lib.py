"""ctypes binding of libchbin_hip.so (C ABI: include/chbin_hip.h).

Fails loudly: if the shared library has not been built, or no MI355X is visible, every compute
call raises -- there is no CPU fallback in this package.
"""
import ctypes as C
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CHBIN_LIB", os.path.join(_HERE, "libchbin_hip.so"))   # CHBIN_LIB: developer override

CHB_MAX_NEIGHBORS = 64

_lib = None
_lock = threading.Lock()
_ctx_by_device = {}


class ChbError(RuntimeError):
    pass


_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")

# every symbol include/chbin_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "chb_last_error": (C.c_char_p, []),
    "chb_version": (C.c_int, []),
    "chb_device_count": (C.c_int, []),
    "chb_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "chb_destroy": (C.c_int, [C.c_void_p]),
    "chb_set_metric": (C.c_int, [C.c_void_p, C.c_int]),
    "chb_set_samples": (C.c_int, [C.c_void_p, _f64p, C.c_int64, C.c_int64]),
    "chb_set_samples_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "chb_pairwise_distance": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, _f64p]),
    "chb_topm_per_bin": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_int, _i64p, C.c_int64, _i64p,
                                   C.c_void_p, _i32p]),
    "chb_find_nearest_from_row": (C.c_int, [C.c_void_p, C.c_int64, _i64p, _f64p, C.c_int64, C.c_int,
                                            _i64p, C.POINTER(C.c_int32)]),
    "chb_hull_distance_batch": (C.c_int, [C.c_void_p, _i64p, C.c_int64, _i64p, C.c_int, _f64p,
                                          C.c_void_p]),
    "chb_hull_distance_points": (C.c_int, [C.c_void_p, _f64p, _f64p, C.c_int, C.c_int64,
                                           C.POINTER(C.c_double), C.c_void_p]),
    "chb_fit_cluster": (C.c_int, [C.c_void_p, C.c_int64, _i64p, _i64p, C.c_int64, C.c_int, C.c_int,
                                  C.c_int, _i64p, C.POINTER(C.c_int), _i64p, C.c_void_p]),
    "chb_fit_cluster_ex": (C.c_int, [C.c_void_p, C.c_int64, _i64p, _i64p, C.c_int64, C.c_int, C.c_int,
                                     C.c_int, _i64p, C.POINTER(C.c_int), _i64p, C.c_void_p, C.c_void_p]),
    "chb_fit_begin": (C.c_int, [C.c_void_p, C.c_int64, _i64p, C.c_int]),
    "chb_batch_begin": (C.c_int, [C.c_void_p, _i64p, C.c_int64, C.c_int64, C.c_int64]),
    "chb_batch_guess": (C.c_int, [C.c_void_p, _i64p]),
    "chb_batch_round": (C.c_int, [C.c_void_p, _i64p, C.c_int64, _i64p, C.c_void_p]),
    "chb_batch_commit": (C.c_int, [C.c_void_p, _i64p]),
    "chb_fit_labels": (C.c_int, [C.c_void_p, _i64p]),
    "chb_comm_unique_id": (C.c_int, [C.c_char_p]),
    "chb_comm_init": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int, C.c_int]),
    "chb_comm_destroy": (C.c_int, [C.c_void_p]),
    "chb_comm_init_hook": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "chb_bcast_samples": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int]),
    "chb_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                C.POINTER(C.c_int)]),
    "chb_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "chb_profile_reset": (C.c_int, [C.c_void_p]),
    "chb_profile_get": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double),
                                  C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "chb_fit_stats": (C.c_int, [C.c_void_p, _i64p]),
    "chb_counter": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "chb_kmer_dim": (C.c_int, [C.c_int]),
    "chb_kmer_frequencies": (C.c_int, [C.c_void_p, C.c_char_p, _i64p, C.c_int64, C.c_int, _f64p, C.c_void_p]),
}


def load():
    """dlopen the HIP library and declare every entry point (no GPU needed for this)."""
    global _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ChbError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                    "g.build()'` (or `make -C ch-bin_amd/csrc`). There is no CPU fallback.")
            lib = C.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)
                fn.restype = res
                fn.argtypes = args
            _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        msg = load().chb_last_error()
        raise ChbError(f"libchbin_hip error {rc}: {msg.decode() if msg else '?'}")


class Context:
    """One libchbin_hip context (one GPU).  Holds the resident feature matrix."""

    def __init__(self, device=0):
        lib = load()
        if lib.chb_device_count() <= 0:
            raise ChbError("no HIP device visible: chbin_amd needs an MI355X (no CPU fallback)")
        self._h = C.c_void_p()
        check(lib.chb_create(int(device), C.byref(self._h)))
        self._lib = lib
        self.device = int(device)
        self.N = self.D = 0
        self._metric = "convex"

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.chb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    METRICS = {"convex": 0, "affine": 1, "affine-qp": 1}

    def set_metric(self, metric: str):
        if metric not in self.METRICS:
            raise NotImplementedError(f"Metric {metric} not implemented")   # hull_distance.py:108
        check(self._lib.chb_set_metric(self._h, self.METRICS[metric]))
        self._metric = metric

    def get_metric(self) -> str:
        return self._metric

    def using_metric(self, metric: str):
        """with ctx.using_metric("affine"): ... -- selects the metric and puts the caller's back afterwards
        (the mirror functions share one default context)."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            before = self._metric
            self.set_metric(metric)
            try:
                yield self
            finally:
                self.set_metric(before)
        return scope()

    # ---- samples
    def set_samples(self, X):
        X = np.ascontiguousarray(X, dtype=np.float64)  # DataFrame.values may be F-ordered
        if X.ndim != 2:
            raise ValueError("samples must be a 2-D array")
        check(self._lib.chb_set_samples(self._h, X, X.shape[0], X.shape[1]))
        self.N, self.D = X.shape

    def set_samples_cached(self, X):
        """Kept for the callers' sake; it ALWAYS uploads.  (It used to skip the upload when the
        array's address / shape / dtype matched the previous call, but the mirror functions hand it
        `np.ascontiguousarray(...)` temporaries whose address is recycled by the next call, and a
        caller may also change `samples` in place: both made the GPU work on stale data.  An upload
        is milliseconds next to a sweep.)"""
        self.set_samples(X)

    def set_samples_device(self, ptr, N, D):
        check(self._lib.chb_set_samples_device(self._h, C.c_void_p(int(ptr)), int(N), int(D)))
        self.N, self.D = int(N), int(D)

    # ---- entry points
    def pairwise_distance(self, r0=0, r1=None):
        r1 = self.N if r1 is None else r1
        out = np.empty((r1 - r0, self.N), dtype=np.float64)
        check(self._lib.chb_pairwise_distance(self._h, int(r0), int(r1), out))
        return out

    def topm_per_bin(self, labels, B, m, query_idx):
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        q = np.ascontiguousarray(query_idx, dtype=np.int64)
        Q = q.shape[0]
        idx = np.empty((Q, B, m), dtype=np.int64)
        dist = np.empty((Q, B, m), dtype=np.float64)
        cnt = np.empty((Q, B), dtype=np.int32)
        check(self._lib.chb_topm_per_bin(self._h, labels, int(B), int(m), q, Q, idx,
                                         dist.ctypes.data, cnt))
        return idx, dist, cnt

    def find_nearest_from_row(self, c, labels, row, m):
        labels = np.ascontiguousarray(labels, dtype=np.int64)
        row = np.ascontiguousarray(row, dtype=np.float64)
        out = np.empty(max(int(m), 1), dtype=np.int64)
        cnt = C.c_int32(0)
        check(self._lib.chb_find_nearest_from_row(self._h, int(c), labels, row, labels.shape[0],
                                                  int(m), out, C.byref(cnt)))
        return out[: cnt.value].copy()

    def hull_distance_batch(self, query_idx, hull_idx, want_alpha=False):
        q = np.ascontiguousarray(query_idx, dtype=np.int64)
        hx = np.ascontiguousarray(hull_idx, dtype=np.int64)
        P, m_max = hx.shape
        dist = np.empty(P, dtype=np.float64)
        alpha = np.zeros((P, m_max), dtype=np.float64) if want_alpha else None
        check(self._lib.chb_hull_distance_batch(self._h, q, P, hx, int(m_max), dist,
                                                None if alpha is None else alpha.ctypes.data))
        return (dist, alpha) if want_alpha else dist

    def hull_distance_points(self, x, pts, want_alpha=False):
        x = np.ascontiguousarray(x, dtype=np.float64)
        pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, x.shape[0])
        m = pts.shape[0]
        d = C.c_double(0.0)
        alpha = np.zeros(max(m, 1), dtype=np.float64)
        if m == 0:
            pts = np.zeros((1, x.shape[0]))
        check(self._lib.chb_hull_distance_points(self._h, x, pts, int(m), x.shape[0], C.byref(d),
                                                 alpha.ctypes.data))
        return (d.value, alpha[:m]) if want_alpha else d.value

    def fit_cluster(self, B, initial_bins, perms, m, max_iter, batch=0, want_min_dist=False):
        initial = np.ascontiguousarray(initial_bins, dtype=np.int64)
        perms = np.ascontiguousarray(perms, dtype=np.int64).reshape(max_iter, -1) \
            if max_iter > 0 else np.zeros((0, 0), dtype=np.int64)
        n_move = perms.shape[1] if max_iter > 0 else 0
        out = np.empty(self.N, dtype=np.int64)
        changed = np.zeros(max(max_iter, 1), dtype=np.int64)
        iters = C.c_int(0)
        mind = np.empty(self.N, dtype=np.float64) if want_min_dist else None
        perms_arg = perms if perms.size else np.zeros(1, dtype=np.int64)
        check(self._lib.chb_fit_cluster(self._h, int(B), initial, perms_arg, int(n_move), int(m),
                                        int(max_iter), int(batch), out, C.byref(iters), changed,
                                        None if mind is None else mind.ctypes.data))
        res = (out, iters.value, changed[: iters.value])
        return res + (mind,) if want_min_dist else res

    def fit_cluster_margins(self, B, initial_bins, perms, m, max_iter, batch=0):
        """fit_cluster plus, per movable contig, runner-up minus winning hull distance at its last visit."""
        initial = np.ascontiguousarray(initial_bins, dtype=np.int64)
        perms = np.ascontiguousarray(perms, dtype=np.int64).reshape(max_iter, -1)
        out = np.empty(self.N, dtype=np.int64)
        changed = np.zeros(max(max_iter, 1), dtype=np.int64)
        iters = C.c_int(0)
        mind = np.empty(self.N, dtype=np.float64)
        margin = np.empty(self.N, dtype=np.float64)
        check(self._lib.chb_fit_cluster_ex(self._h, int(B), initial, perms, int(perms.shape[1]), int(m),
                                           int(max_iter), int(batch), out, C.byref(iters), changed,
                                           mind.ctypes.data, margin.ctypes.data))
        return out, iters.value, margin

    # stepwise (multi-GPU driver)
    def fit_begin(self, B, initial_bins, m):
        check(self._lib.chb_fit_begin(self._h, int(B), np.ascontiguousarray(initial_bins, dtype=np.int64), int(m)))

    def batch_begin(self, perm_slice, q_lo, q_hi):
        p = np.ascontiguousarray(perm_slice, dtype=np.int64)
        check(self._lib.chb_batch_begin(self._h, p, p.shape[0], int(q_lo), int(q_hi)))

    def batch_guess(self, guess):
        check(self._lib.chb_batch_guess(self._h, guess))

    def batch_round(self, lab_prev, active, lab_new, min_dist=None):
        check(self._lib.chb_batch_round(self._h, np.ascontiguousarray(lab_prev, dtype=np.int64),
                                        int(active), lab_new,
                                        None if min_dist is None else min_dist.ctypes.data))

    def batch_commit(self, final_labels):
        check(self._lib.chb_batch_commit(self._h, np.ascontiguousarray(final_labels, dtype=np.int64)))

    def fit_labels(self):
        out = np.empty(self.N, dtype=np.int64)
        check(self._lib.chb_fit_labels(self._h, out))
        return out

    # multi-GPU (RCCL inside chb_fit_cluster)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(load().chb_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        check(self._lib.chb_comm_init(self._h, C.c_char_p(unique_id), int(rank), int(world)))
        self.rank, self.world = int(rank), int(world)

    ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)

    def comm_init_hook(self, rank: int, world: int, allgather):
        """The sharded C++ loop with a host-side exchange: allgather(send: np.uint8[bytes]) ->
        np.uint8[world * bytes] (rank-major)."""
        def _cb(_user, send, recv, nbytes):
            try:
                src = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_uint8)), shape=(nbytes,))
                out = np.ascontiguousarray(allgather(src.copy()), dtype=np.uint8).reshape(-1)
                if out.size != nbytes * world:
                    return 1
                C.memmove(recv, out.ctypes.data, out.size)
                return 0
            except Exception:  # noqa: BLE001
                return 2
        self._hook_cb = self.ALLGATHER_FN(_cb)      # keep the trampoline alive
        check(self._lib.chb_comm_init_hook(self._h, int(rank), int(world), C.cast(self._hook_cb, C.c_void_p), None))
        self.rank, self.world = int(rank), int(world)

    def comm_destroy(self):
        check(self._lib.chb_comm_destroy(self._h))

    def bcast_samples(self, X, N, D, root=0):
        """chb_set_samples for all ranks of the RCCL communicator: the root passes its host matrix, the others None;
        the matrix crosses the host boundary once and reaches the other GPUs by an RCCL broadcast over xGMI."""
        ptr = None
        if X is not None:
            X = np.ascontiguousarray(X, dtype=np.float64)
            if X.shape != (int(N), int(D)):
                raise ValueError("samples must be an N x D matrix")
            ptr = X.ctypes.data_as(C.c_void_p)
        check(self._lib.chb_bcast_samples(self._h, ptr, int(N), int(D), int(root)))
        self.N, self.D = int(N), int(D)

    def comm_info(self):
        """{'rank', 'world', 'comm_ranks' (what RCCL reports for the communicator; 0 without one), 'transport'}"""
        r, w, n, t = C.c_int(0), C.c_int(1), C.c_int(0), C.c_int(0)
        check(self._lib.chb_comm_info(self._h, C.byref(r), C.byref(w), C.byref(n), C.byref(t)))
        return {"rank": r.value, "world": w.value, "comm_ranks": n.value,
                "transport": {0: "none", 1: "rccl", 2: "hook"}[t.value]}

    # measurement
    def profile_enable(self, on=True):
        """on: False / 0 off, True / 1 every kernel, 2 only the two dominant kernels."""
        check(self._lib.chb_profile_enable(self._h, int(on)))

    def profile_reset(self):
        check(self._lib.chb_profile_reset(self._h))

    def profile_get(self, kernel):
        ms, n, w = C.c_double(0), C.c_int64(0), C.c_double(0)
        check(self._lib.chb_profile_get(self._h, kernel.encode(), C.byref(ms), C.byref(n), C.byref(w)))
        return {"ms": ms.value, "launches": n.value, "work": w.value}

    def counter(self, name):
        v = C.c_int64(0)
        check(self._lib.chb_counter(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def kmer_frequencies(self, sequences, k=4, return_counts=False):
        """chb_kmer_frequencies: sequences = list of bytes / str, one per contig."""
        bs = [s.encode() if isinstance(s, str) else bytes(s) for s in sequences]
        dim = self._lib.chb_kmer_dim(int(k))
        if dim <= 0:
            raise ChbError(self._lib.chb_last_error().decode())
        offsets = np.zeros(len(bs) + 1, dtype=np.int64)
        if bs:
            offsets[1:] = np.cumsum([len(b) for b in bs])
        freq = np.zeros((len(bs), dim), dtype=np.float64)
        counts = np.zeros((len(bs), dim), dtype=np.uint32) if return_counts else None
        check(self._lib.chb_kmer_frequencies(self._h, b"".join(bs), offsets, len(bs), int(k), freq.reshape(-1),
                                              counts.ctypes.data_as(C.c_void_p) if counts is not None else None))
        return (freq, counts) if return_counts else freq

    def fit_stats(self):
        out = np.zeros(4, dtype=np.int64)
        check(self._lib.chb_fit_stats(self._h, out))
        return {"batches": int(out[0]), "rounds": int(out[1]), "hull_evaluated": int(out[2]),
                "hull_needed": int(out[3])}


def default_context(device=None):
    """Process-wide context for `device` (default: LOCAL_RANK or 0)."""
    if device is None:
        device = int(os.environ.get("CHBIN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    with _lock:
        ctx = _ctx_by_device.get(device)
    if ctx is None:
        ctx = Context(device)
        with _lock:
            _ctx_by_device[device] = ctx
    return ctx
