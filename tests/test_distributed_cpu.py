"""CPU coverage of the speculative-batch control flow and of the N>1 rank protocol
(chbin_amd.distributed) with the oracle standing in for the per-rank HIP evaluation."""
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import chbin_amd
from chbin_amd import distributed as D
from oracle import oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_backend import OracleBackend  # noqa: E402


def _case(seed=4, N=260, Dm=12, B=4, mix=0.6):
    X, initial, _ = chbin_amd.synth.make_synthetic(N, Dm, B, seed=seed, sigma=8e-3, mix=mix, n_seed=5)
    perms = chbin_amd.synth.draw_permutations(initial, 5, seed=0)
    return X, initial, perms, B


def test_slice_bounds_cover():
    for K in (1, 7, 64, 4096):
        for w in (1, 2, 3, 8):
            b = [D.slice_bounds(K, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == K
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))


def test_batch_schedule_partitions():
    for n, batch, mem, first in ((1000, 128, 10, True), (1000, 0, 5000, False), (5, 64, 0, True)):
        s = D.batch_schedule(n, batch, mem, first)
        assert s[0][0] == 0 and sum(k for _, k in s) == n
        assert all(s[i][0] + s[i][1] == s[i + 1][0] for i in range(len(s) - 1))


@pytest.mark.parametrize("world,batch", [(1, 1), (1, 50), (3, 64), (2, 4096)])
def test_speculative_rounds_equal_sequential(world, batch):
    """Any batch size / any number of slices reproduces the sequential Gauss-Seidel sweep."""
    X, initial, perms, B = _case()
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, 3, 5)
    got, its, ch = D.run_sweeps([OracleBackend() for _ in range(world)], X, B, initial, perms, 3, 5,
                                batch=batch)
    assert its == its_o and np.array_equal(ch, ch_o)
    assert np.array_equal(got, want)
    assert ch[0] > 0 and len(ch) >= 2     # the case really moves contigs over several sweeps


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        X, initial, perms, B = _case(seed=8)
        labels, its, ch = D.fit_cluster_distributed(OracleBackend(), X, B, initial, perms, 3, 5, batch=40)
        q.put((rank, labels, its, ch))
    finally:
        dist.destroy_process_group()


def test_gloo_world_size_2():
    X, initial, perms, B = _case(seed=8)
    want, its_o, ch_o = O.fit_cluster(X, B, initial, perms, 3, 5)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, labels, its, ch in res:
        assert its == its_o and np.array_equal(ch, ch_o)
        assert np.array_equal(labels, want)


def test_bench_self_launch_prints_one_line():
    """`python bench.py --gpus 2` with no launcher in the environment starts its two ranks itself (as children, before
    anything touches a GPU), relays exactly one JSON line from rank 0 and returns the launcher's status.  --launch-check
    stops the ranks after their first collective (gloo), so this runs without a GPU."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"launch_check": True, "world": 2, "rank_sum": 3.0}
    # a launcher that started the wrong number of ranks is an error, not a silent single-GPU run
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(env, WORLD_SIZE="1", RANK="0"),
                       timeout=300)
    assert p.returncode != 0 and p.stdout == ""
