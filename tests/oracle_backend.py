"""A stepwise backend (same methods as chbin_amd._lib.Context) implemented with the CPU ORACLE.
Test infrastructure: lets the CPU suite exercise the speculative-batch / multi-rank control flow of
chbin_amd.distributed without a GPU.  Never used by the product."""
import numpy as np

from oracle import oracle as O


class OracleBackend:
    def set_samples(self, X):
        self.X = np.ascontiguousarray(X, dtype=np.float64)

    def fit_begin(self, B, initial, m):
        self.B, self.m = int(B), int(m)
        self.labels = np.ascontiguousarray(initial, dtype=np.int64).copy()

    def batch_begin(self, perm_slice, q_lo, q_hi):
        self.sl = np.asarray(perm_slice, dtype=np.int64).copy()
        self.lo, self.hi = int(q_lo), int(q_hi)
        self.lab_old = self.labels[self.sl].copy()

    def batch_round(self, lab_prev, active, lab_new, min_dist=None):
        lab_prev = np.asarray(lab_prev, dtype=np.int64)
        for pos in range(max(self.lo, int(active)), self.hi):
            # the label state the sequential loop would see when it visits position `pos`:
            # earlier batch members carry their (speculative) new label, later ones their old one
            tmp = self.labels.copy()
            tmp[self.sl[:pos]] = lab_prev[:pos]
            tmp[self.sl[pos:]] = self.lab_old[pos:]
            j = self.sl[pos]
            out, md = O.sweep(self.X, self.B, tmp, np.array([j]), self.m)
            lab_new[pos] = out[j]
            if min_dist is not None:
                min_dist[pos] = md[0]

    def batch_commit(self, final):
        self.labels[self.sl] = np.asarray(final, dtype=np.int64)
