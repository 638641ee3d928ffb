"""Multi-GPU form of fit_cluster: one process per GPU, contigs of every speculative batch sharded
across ranks.

The reference has nothing distributed (single process, single thread).  What shards here is the
batch: the K (contig) x B (bin) hull distances of a speculative round are independent, so rank r
evaluates the contiguous slice [q_lo, q_hi) of the batch's positions against its own replica of
the feature matrix and labels.  The only exchange is the K new labels (+K winning distances) per
round -- KB-sized, latency-bound -- done with one all_reduce(SUM) over disjoint slices (RCCL on
GPUs, gloo in the CPU tests).  Every rank then applies the same deterministic commit, so labels
stay replicated without further traffic.  Results are identical to the single-GPU (and the
reference's sequential) result for any world size and any batch size.

A backend is anything with the stepwise methods of `_lib.Context`:
    set_samples(X), fit_begin(B, initial, m), batch_begin(perm_slice, q_lo, q_hi),
    batch_round(lab_prev, active, lab_new, min_dist), batch_commit(final)
"""
import numpy as np


def slice_bounds(K, world, rank):
    """Contiguous, balanced split of batch positions [0, K) over `world` ranks."""
    base, rem = divmod(int(K), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def batch_schedule(n_move, batch, members0, first_sweep):
    """Batch sizes for one sweep.  Any partition gives the same labels; in sweep 1 a batch is not
    allowed to outnumber the members labelled so far by more than half (keeps speculation rounds few)."""
    out, t0 = [], 0
    kmax = max(1, min(int(batch) if batch and batch > 0 else 8192, max(n_move, 1)))
    while t0 < n_move:
        members = (members0 + t0) * 3 // 2 if first_sweep else 1 << 62   # (as csrc/chb_api.hip)
        K = min(kmax, n_move - t0)
        if members < K:
            K = min(max(min(64, n_move - t0), members), kmax)
        out.append((t0, K))
        t0 += K
    return out


def _sweeps(n, B, initial, perms, max_iter, batch, open_batch, evaluate, commit, guess=None):
    """Shared control flow (mirrors csrc/chb_api.hip:chb_fit_cluster)."""
    initial = np.ascontiguousarray(initial, dtype=np.int64)
    perms = np.ascontiguousarray(perms, dtype=np.int64).reshape(max_iter, -1)
    n_move = perms.shape[1]
    labels = initial.copy()
    prev = initial.copy()
    members0 = int((initial >= 0).sum())
    changed = []
    its = 0
    for it in range(max_iter):
        for t0, K in batch_schedule(n_move, batch, members0, it == 0):
            sl = perms[it, t0:t0 + K]
            open_batch(sl)
            lab_prev = labels[sl].copy()
            if guess is not None:
                lab_prev = guess(lab_prev)           # only a starting point: any start is exact
            active = 0
            while True:
                lab_new = evaluate(lab_prev, active)
                diff = np.flatnonzero(lab_new[active:] != lab_prev[active:])
                lab_prev[active:] = lab_new[active:]
                if diff.size == 0:
                    break
                active = active + int(diff[0]) + 1   # positions <= first change are final
                if active >= K:
                    break
            commit(lab_prev)
            labels[sl] = lab_prev
        its = it + 1
        d = int((prev != labels).sum())                # algorithm.py:63-68
        changed.append(d)
        if d == 0:
            break
        prev = labels.copy()
    return labels, its, np.asarray(changed, dtype=np.int64)


def run_sweeps(backends, X, B, initial, perms, m, max_iter, batch=0):
    """All ranks emulated in ONE process: backends[r] plays rank r (used by tests; also a handy
    way to drive several GPUs from one host thread)."""
    world = len(backends)
    X = np.ascontiguousarray(X, dtype=np.float64)
    for b in backends:
        b.set_samples(X)
        b.fit_begin(int(B), initial, int(m))
    state = {}

    def open_batch(sl):
        state["K"] = len(sl)
        for r, b in enumerate(backends):
            lo, hi = slice_bounds(len(sl), world, r)
            b.batch_begin(sl, lo, hi)

    def evaluate(lab_prev, active):
        K = state["K"]
        merged = lab_prev.copy()
        for r, b in enumerate(backends):
            lo, hi = slice_bounds(K, world, r)
            out = lab_prev.copy()
            b.batch_round(lab_prev, active, out, None)
            lo = max(lo, active)
            merged[lo:hi] = out[lo:hi]
        return merged

    def commit(final):
        for b in backends:
            b.batch_commit(final)

    def guess(lab_old):
        g = lab_old.copy()
        for b in backends:
            if hasattr(b, "batch_guess"):
                b.batch_guess(g)                 # each backend fills its own slice
        return g

    return _sweeps(len(X), B, initial, perms, max_iter, batch, open_batch, evaluate, commit, guess)


def fit_cluster_distributed(backend, X, B, initial, perms, m, max_iter, batch=0, group=None,
                            device=None):
    """One rank of the N-process job (torch.distributed already initialised; backend 'nccl' is
    RCCL over xGMI on MI355X, 'gloo' on CPU).  Every rank passes the same X / initial / perms and
    gets the same labels back."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    X = np.ascontiguousarray(X, dtype=np.float64)
    getattr(backend, "set_samples_cached", backend.set_samples)(X)   # resident copy is reused
    backend.fit_begin(int(B), initial, int(m))
    state = {}

    def open_batch(sl):
        state["K"] = len(sl)
        lo, hi = slice_bounds(len(sl), world, rank)
        backend.batch_begin(sl, lo, hi)

    def evaluate(lab_prev, active):
        K = state["K"]
        lo, hi = slice_bounds(K, world, rank)
        out = lab_prev.copy()
        backend.batch_round(lab_prev, active, out, None)
        merged = _merge(out, max(lo, active), hi)
        merged[:active] = lab_prev[:active]
        return merged

    def commit(final):
        backend.batch_commit(final)

    def _merge(arr, lo, hi):
        contrib = np.zeros(len(arr), dtype=np.int64)
        if hi > lo:
            contrib[lo:hi] = arr[lo:hi] + 1          # shift so that "not mine" == 0
        t = torch.from_numpy(contrib)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy() - 1

    def guess(lab_old):
        if not hasattr(backend, "batch_guess"):
            return lab_old
        g = lab_old.copy()
        backend.batch_guess(g)
        lo, hi = slice_bounds(state["K"], world, rank)
        return _merge(g, lo, hi)

    labels, its, changed = _sweeps(len(X), B, initial, perms, max_iter, batch, open_batch, evaluate,
                                   commit, guess)
    # algorithm.py:63-69 convergence statistics are identical on every rank by construction;
    # one tiny all_reduce(MAX) asserts the replicas did not diverge.
    chk = torch.tensor([int(labels.sum()), int(its)], dtype=torch.int64)
    chk_max = chk.clone()
    if device is not None:
        chk_max = chk_max.to(device)
    dist.all_reduce(chk_max, op=dist.ReduceOp.MAX, group=group)
    if not torch.equal(chk_max.cpu(), chk):
        raise RuntimeError("label replicas diverged across ranks")
    return labels, its, changed


def init_native_comm(ctx, group=None, device=None):
    """Create the RCCL communicator that lets `ctx.fit_cluster` shard batches across all ranks in
    C++ (csrc/chb_api.hip).  torch.distributed is only used to hand rank 0's 128-byte unique id to
    the other ranks.  Afterwards every rank calls ctx.fit_cluster(...) with identical arguments."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    buf = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        buf = torch.frombuffer(bytearray(type(ctx).comm_unique_id()), dtype=torch.uint8).clone()
    if device is not None:
        buf = buf.to(device)
    dist.broadcast(buf, src=0, group=group)
    ctx.comm_init(bytes(buf.cpu().numpy().tobytes()), rank, world)
    return rank, world
