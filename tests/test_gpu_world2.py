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
    (1200, 300, 6, 5, 3, 3e-3, 0.5, 10, 300),      # wide rows (round 5: three 144-column slices) on position slices that do not start at 0
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


# ---------------------------------------------------------------------------------------------------------------------
# The ORDER of the exchanges under the look-ahead (round 5).  Under the RCCL transport a rank that looks ahead enqueues the
# next batch's all-gathers before the current batch's verdict; whether it does depends on the tile-skipping verdict and on
# the persistent pack's state, which up to round 4 every rank derived from its OWN launches' statistics: ranks on different
# sides of the threshold issued their collectives in different orders (mixed label buffers, a hang at the sweep's end).
# Now the statistics travel in the frames of the label all-gather and every rank decides from their sum.  RCCL cannot run
# two ranks on one GPU, so the developer library lets the host-staged hook run the look-ahead path (CHB_DEV_HOOK_SPEC=1:
# same order of exchanges, speculative ones included) and replaces a rank's statistics (CHB_DEV_SKIP_STATS), and the hook
# of this test checks that all ranks are in the same exchange (equal byte counts) before it moves data.
SCHED_WORKER = r"""
import os, sys, datetime
import numpy as np
sys.path.insert(0, sys.argv[1])
rank, world, port, out = int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
stats = sys.argv[6].split(";")
os.environ["CHB_DEV_SKIP_STATS"] = stats[min(rank, len(stats) - 1)]
import torch
import torch.distributed as dist
import chbin_amd
from chbin_amd import _lib, synth
dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world,
                        timeout=datetime.timedelta(seconds=25))
calls = [0]

def allgather(send):
    calls[0] += 1
    n = torch.tensor([send.size, calls[0]], dtype=torch.int64)
    ns = [torch.empty_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    if any(int(v[0]) != send.size for v in ns):
        raise RuntimeError("ranks in different exchanges")
    t = torch.from_numpy(send)
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return torch.cat(outs).numpy()

res = {"error": ""}
try:
    N, D, B, S, m, iters, batch, n_seed = (int(v) for v in sys.argv[7].split(","))
    sigma, mix = (float(v) for v in sys.argv[8].split(","))
    X, initial, _ = synth.make_synthetic(N, D, B, S=S, seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = synth.draw_permutations(initial, iters, seed=0)
    ctx = _lib.Context(0)
    ctx.comm_init_hook(rank, world, allgather)
    ctx.set_samples(X)
    lab, its, ch = ctx.fit_cluster(B, initial, perms, m, iters, batch=batch)
    st = ctx.fit_stats()
    res.update(lab=lab, its=its, batches=st["batches"], rounds=st["rounds"], lookahead=ctx.counter("lookahead_batches"),
               skip_state=ctx.counter("tile_skip_state"), calls=calls[0], failed=ctx.counter("lookahead_failed"),
               exchanges=ctx.counter("exchanges"))
    ctx.comm_destroy()
    ctx.close()
except Exception as e:  # noqa: BLE001
    res["error"] = repr(e)
np.savez(out, **res)
os._exit(0)     # (after a failed exchange the other rank may sit in a collective: no barrier, no destructor)
"""


# bins that overlap enough for some batches to need further rounds (failed look-aheads) while others converge at once
SCHED_DATA = ((3000, 136, 8, 1, 5, 3, 32, 8), (4.5e-3, 0.4))


def _run_sched(world, stats, extra_env, data=SCHED_DATA):
    root_dev = os.path.join(ROOT, "ch-bin_amd", "libchbin_hip_dev.so")
    if not os.path.exists(root_dev):
        pytest.skip("developer library not built")
    with tempfile.TemporaryDirectory() as td:
        script = os.path.join(td, "worker.py")
        open(script, "w").write(SCHED_WORKER)
        port = str(31500 + (os.getpid() + 13 * world + len(stats)) % 2000)
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", CHBIN_LIB=root_dev, CHB_DEV_HOOK_SPEC="1", **extra_env)
        procs = [subprocess.Popen([sys.executable, script, ROOT, str(r), str(world), port, os.path.join(td, f"o{r}.npz"), stats,
                                   ",".join(str(v) for v in data[0]), ",".join(str(v) for v in data[1])], env=env)
                 for r in range(world)]
        for p in procs:
            p.wait(timeout=600)
        return [dict(np.load(os.path.join(td, f"o{r}.npz"))) if os.path.exists(os.path.join(td, f"o{r}.npz")) else None
                for r in range(world)]


def _sched_oracle():
    from oracle import oracle as O
    import chbin_amd
    (N, D, B, S, m, iters, batch, n_seed), (sigma, mix) = SCHED_DATA
    X, initial, _ = chbin_amd.synth.make_synthetic(N, D, B, S=S, seed=N + B, sigma=sigma, mix=mix, n_seed=n_seed)
    perms = chbin_amd.synth.draw_permutations(initial, iters, seed=0)
    return oracle_fit_replay(O, X, B, initial, perms, m, iters, key=("sched", 0))


# rank 0 reports that its launch skipped everything, the others that theirs skipped nothing / (second case) nobody skips,
# and rank 1's launches report no wave-tile at all (an empty shard counts its batches later)
@pytest.mark.parametrize("world,stats", [(2, "1000,1000,1000;0,1000,0"), (3, "1000,1000,1000;0,1000,0"),
                                         (2, "0,1000,0;0,0,0"), (3, "0,1000,0;0,0,0;0,2000,0")])
def test_lookahead_schedule_is_the_same_on_every_rank(world, stats):
    outs = _run_sched(world, stats, {})
    want, its_o, _, _ = _sched_oracle()
    for r, o in enumerate(outs):
        assert o is not None and str(o["error"]) == "", (r, o and str(o["error"]))
        assert int(o["its"]) == its_o and np.array_equal(o["lab"], want)
        # look-aheads were kept AND failed (batches with more than one round), so speculative exchanges did go out
        assert int(o["lookahead"]) > 0 and int(o["failed"]) > 0 and int(o["rounds"]) > int(o["batches"])
        # (every speculative exchange of a discarded look-ahead went out as well: more exchanges than batches + rounds)
        assert int(o["exchanges"]) >= int(o["batches"]) + int(o["rounds"]) + 2 * int(o["failed"])
    for key in ("batches", "rounds", "lookahead", "failed", "skip_state", "calls", "exchanges"):
        assert len({int(o[key]) for o in outs}) == 1, key                 # one schedule, one verdict
    assert int(outs[0]["skip_state"]) == (1 if stats.startswith("1000") else -1)


def test_per_rank_verdicts_do_break_the_schedule_and_are_caught():
    """The behaviour before round 5 (CHB_DEV_LOCAL_VERDICT=1: every rank decides from its own statistics) on the same
    inputs: the ranks end up in different exchanges -- the self-checking hook or the frames' tags notice it and the fit
    FAILS on the ranks instead of returning mixed labels.  (What the RCCL transport would have done here is undefined.)"""
    # (CHB_PACK_REBUILD_AT=1: the rank that turned skipping off rebuilds its persistent pack at every batch start and so
    #  never looks ahead -- the divergence of ONE batch that per-rank verdicts cause, made permanent)
    outs = _run_sched(2, "1000,1000,1000;0,1000,0", {"CHB_DEV_LOCAL_VERDICT": "1", "CHB_PACK_REBUILD_AT": "1"})
    errs = [None if o is None else str(o["error"]) for o in outs]
    assert any(e is None or e != "" for e in errs), errs
    assert any(e and ("out of step" in e or "exchange hook" in e) for e in errs), errs
