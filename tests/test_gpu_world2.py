"""World-size-2 and -3 runs of the SHARDED C++ loop (chb_fit_cluster with q_lo/q_hi slices and all-gathers between
rounds, csrc/chb_api.hip) on one GPU: one process and one context per rank, exchange through the host-staged hook
(chb_comm_init_hook) carried by gloo.  RCCL refuses two ranks on one device, so this is how the loop that the
driver runs over RCCL on 8 GPUs is exercised with more than one rank before it gets there: same slices, same
exchange points, same first-change logic -- only the transport differs."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle_backend import oracle_fit_replay  # noqa: E402

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
import torch
import torch.distributed as dist
import chbin_amd
from chbin_amd import _lib, synth
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)

def allgather(send):                      # np.uint8[bytes] -> np.uint8[world * bytes]
    t = torch.from_numpy(send)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.cat(outs).numpy()

CASES = [
    (900, 136, 8, 5, 4, 6e-3, 0.6, 8, 200),        # overlapping bins: several rounds per batch
    (3000, 136, 16, 5, 3, 1.5e-3, 0.0, None, 0),   # SURVEY 8(d) generator
    (700, 40, 6, 15, 3, 9e-3, 0.5, 20, 150),       # m = 15: fused 16-lane kernel (hull_select_qp16_kernel)
    (10000, 136, 32, 5, 4, 1.5e-3, 0.0, None, 0),  # BASELINE configs[1] at its stated size (seed 0 as in test_gpu_configs)
    (6000, 140, 12, 5, 3, 2e-3, 0.2, 60, 0),       # five coverage columns: the tile-skipping shortlist build on a rank's slice
]
res = {}
for case, (N, D, B, m, iters, sigma, mix, n_seed, batch) in enumerate(CASES):
    seed = 0 if N == 10000 else N + B
    X, initial, _ = synth.make_synthetic(N, D, B, S=5 if D == 140 else 1, seed=seed, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = synth.draw_permutations(initial, iters, seed=0)
    ctx = _lib.Context(0)
    ctx.comm_init_hook(rank, world, allgather)
    assert ctx.comm_info() == {"rank": rank, "world": world, "comm_ranks": 0, "transport": "hook"}
    ctx.set_samples(X)
    lab, its, ch, mind = ctx.fit_cluster(B, initial, perms, m, iters, batch=batch, want_min_dist=True)
    st = ctx.fit_stats()
    res[f"lab{case}"] = lab; res[f"its{case}"] = its; res[f"mind{case}"] = mind
    res[f"evaluated{case}"] = st["hull_evaluated"]; res[f"needed{case}"] = st["hull_needed"]
    res[f"rounds{case}"] = st["rounds"]
    res[f"unloaded{case}"] = ctx.counter("tile_unloaded")
    ctx.comm_destroy()
    ctx.close()
np.savez(out, **res)
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_cpp_loop_more_than_one_rank_one_gpu(world):
    from oracle import oracle as O
    import chbin_amd
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "worker.py")
        open(script, "w").write(WORKER)
        port = str(29500 + (os.getpid() + 7 * world) % 2000)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), str(world), port,
                                   os.path.join(td, f"out{r}.npz")], env=env) for r in range(world)]
        for p in procs:
            assert p.wait(timeout=900) == 0
        outs = [np.load(os.path.join(td, f"out{r}.npz")) for r in range(world)]
    ns = {}
    exec(WORKER[WORKER.index("CASES = ["):WORKER.index("res = {}")], ns)       # the worker's own case table
    for case, (N, D, B, m, iters, sigma, mix, n_seed, batch) in enumerate(ns["CASES"]):
        seed = 0 if N == 10000 else N + B
        X, initial, _ = chbin_amd.synth.make_synthetic(N, D, B, S=5 if D == 140 else 1, seed=seed, sigma=sigma, mix=mix,
                                                       n_seed=n_seed)
        perms = chbin_amd.synth.draw_permutations(initial, iters, seed=0)
        # (one replay of the oracle's sweeps per case, shared by the world-2 and the world-3 run)
        want, its_o, _, md = oracle_fit_replay(O, X, B, initial, perms, m, iters, key=("world", case))
        for r in range(world):
            assert int(outs[r][f"its{case}"]) == its_o
            assert np.array_equal(outs[r][f"lab{case}"], want)                       # every rank: the full result
            assert np.allclose(outs[r][f"mind{case}"][perms[its_o - 1]], md, rtol=0, atol=1e-9)
        # the ranks evaluated DISJOINT slices: together exactly what one rank alone would have evaluated
        ev = [int(outs[r][f"evaluated{case}"]) for r in range(world)]
        needed = int(outs[0][f"needed{case}"])
        assert all(e > 0 for e in ev) and sum(ev) >= needed
        assert max(ev) <= (0.75 if world == 2 else 0.6) * sum(ev)                    # no rank did (nearly) all of it
        assert len({int(outs[r][f"rounds{case}"]) for r in range(world)}) == 1
        if D == 140:   # tiles really were skipped on the ranks' slices (the first workgroups of a launch are sampled)
            assert sum(int(outs[r][f"unloaded{case}"]) for r in range(world)) > 0


def test_bench_self_launch_two_ranks_one_gpu():
    """`python bench.py --gpus 2` with no launcher in the environment, on a box with ONE GPU: the two ranks are started as
    children (before anything touches the GPU), share cuda:0 over gloo and -- RCCL refuses two ranks on one device -- run the
    Python driver; exactly one JSON line comes back and the status is 0.  (The driver's own multi-GPU runs go through RCCL;
    this covers the launch path on a GPU box.)"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--same-gpu",
                        "--allow-fallback", "--contigs", "12000", "--bins", "16", "--steps", "1", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.splitlines()
    assert len(lines) == 1
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0
    ev = j["config"]["parallelism_evidence"]
    assert len(ev["hull_evaluated_per_rank_last_step"]) == 2 and all(v > 0 for v in ev["hull_evaluated_per_rank_last_step"])
