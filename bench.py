#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: convex-hull QP distances/sec (+ end-to-end bin-assign
wall-clock) on a synthetic N-contig x D-dim x B-bin workload; default = BASELINE.json configs[2]
(N=100k, D=136, B=64, AlgoNumNeighbors=5).

A "step" is ONE complete fit_cluster sweep (algorithm.py:43-60) from the seed state: every movable
contig (~98k) is visited in the reference's permutation order and, for each of the B bins, its m
nearest members are selected and the point-to-convex-hull QP distance is evaluated (~6.27M hull
distances), with the reference's sequential label semantics.  The feature matrix is resident in
HBM before the timed region; labels/permutation cross the boundary every step exactly as the
reference's call does.  `value` = hull distances the sequential loop needs / wall time
(speculative re-evaluations are NOT counted as work).

The LAST timed step carries HIP events around every launch of the two dominant kernels (four event records
per batch: the roofline's live launch durations; a pair costs the stream about 8 us, 0.2 ms of a 10 ms step, so the
other timed steps run without them -- `--kernel-events-every-step` keeps them on); the full per-kernel table comes
from a separate, untimed pass of the same steps.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0          # HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
HBM_COPY_PEAK_GBS = 6290.0     # the measured-copy peak BASELINE.md 3 also quotes the path figure against
F16_PEAK_TFLOPS = 2500.0       # dense fp16/bf16 MFMA peak
TRAFFIC_FILE = "r05_traffic.json"   # rocprofv3 --pmc passes of this round (tools/traffic_from_pmc.py); m > 5: r05_m15_traffic.json;
                                    # BASELINE configs[3] / [4]: r05_cfg3_traffic.json / r05_cfg4_traffic.json
# MI355X_MICROARCH.md: indexed rows out of an L2-resident table gather at 16.8-18.8 TB/s chip-wide (mid-point); L2 aggregate 34.5
L2_GATHER_PEAK_GBS = 17800.0
L2_AGGREGATE_GBS = 34500.0


def gather_ceiling_gbs(table_mb):
    """MI355X_MICROARCH.md, 'Indexed rows: gather into LDS', chip-wide, by the size of the table the rows come from:
    2,048 shared rows (2.3 MB, one XCD's L2) 16.8-18.8 TB/s; 38 MB (Infinity Cache) 8.6; 151 MB 7.4-7.9; 1.2 GB swept
    (HBM) 6.0-6.1.  Piecewise-linear in between (mid-points of the quoted ranges)."""
    pts = [(2.3, 17800.0), (38.0, 8600.0), (151.0, 7650.0), (1200.0, 6050.0)]
    if table_mb <= pts[0][0]:
        return pts[0][1]
    for (x0, y0), (x1, y1) in zip(pts, pts[1:]):
        if table_mb <= x1:
            return y0 + (y1 - y0) * (table_mb - x0) / (x1 - x0)
    return pts[-1][1]


def _code_only(text):
    """C++ source without comments, whitespace runs collapsed (string / character literals kept as they are)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c in "\"'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            if out and out[-1] != " ":
                out.append(" ")
            i = n if j < 0 else j + 2
        elif c.isspace():
            if out and out[-1] != " ":
                out.append(" ")
            i += 1
        else:
            out.append(c)
            i += 1
    return "".join(out).strip()


def kernel_source_stamp():
    """sha256 over the kernel sources' CODE (comments and layout do not count): profiles/*_traffic.json carries the stamp
    of the build it was measured on, and its numbers are only quoted while the code is still the same."""
    import glob
    import hashlib
    hh = hashlib.sha256()
    src = os.path.join(ROOT, "ch-bin_amd", "csrc")
    for f in sorted(glob.glob(os.path.join(src, "*.hip")) + glob.glob(os.path.join(src, "*.h"))):
        hh.update(os.path.basename(f).encode())
        hh.update(_code_only(open(f, "r", encoding="utf-8", errors="replace").read()).encode())
    return hh.hexdigest()[:16]


FP64_PEAK_TFLOPS = 78.6        # fp64 vector == matrix peak (AMD public spec; the local guide lists no fp64 row)

# BASELINE.json configs: index -> (contigs, dim, bins)
BASELINE_CONFIGS = {1: (10_000, 136, 32), 2: (100_000, 136, 64), 3: (500_000, 140, 128), 4: (1_000_000, 146, 200)}


def coverage_columns(D):
    return 1 if D <= 136 else (5 if D == 140 else 10)


def config_name(N, D, B):
    for idx, shape in BASELINE_CONFIGS.items():
        if shape == (N, D, B):
            return f"BASELINE configs[{idx}]"
    return "custom (not a BASELINE config)"


def self_launch(n):
    """One rank per GPU as child processes (the driver's own launch line, with a free rendezvous port); this process never
    initialises a GPU and replaces nothing -- it waits, prints rank 0's JSON line and returns the launcher's exit status."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    p = subprocess.run(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    for ln in p.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if p.returncode == 0 and len(lines) != 1:
        print(f"self-launch: expected one result line from rank 0, got {len(lines)}", file=sys.stderr)
        return 4
    if lines:
        print(lines[-1], flush=True)
    return p.returncode


def launch_check(n):
    """--launch-check: what a rank does up to the first collective, without a GPU (tests/test_distributed_cpu.py)."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != n:
        print(f"--gpus {n} but WORLD_SIZE={world}", file=sys.stderr)
        return 2
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)
    dist.barrier()
    if rank == 0:
        print(json.dumps({"launch_check": True, "world": world, "rank_sum": float(t.item())}), flush=True)
    dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--contigs", type=int, default=100_000)
    ap.add_argument("--dim", type=int, default=136)
    ap.add_argument("--bins", type=int, default=64)
    ap.add_argument("--neighbors", type=int, default=5)      # config/default.ini:16
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--mix", type=float, default=0.0, help="synthetic generator: pull of the bins towards a common profile")
    ap.add_argument("--sigma", type=float, default=1.5e-3)
    ap.add_argument("--cpu-sample", type=int, default=300)
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the m = 15 / overlapping-bins / label-margin legs")
    ap.add_argument("--traffic-file", default=None,
                    help="PMC traffic table to quote instead of profiles/%s (tools/collect_profiles.sh: the table it has "
                         "just measured on this very build)" % TRAFFIC_FILE)
    ap.add_argument("--no-kernel-events", action="store_true", help="dev: no HIP events at all in the timed steps")
    ap.add_argument("--kernel-events-every-step", action="store_true",
                    help="HIP events around the two dominant kernels in EVERY timed step instead of the last one only "
                         "(an event pair costs the stream ~8 us per launch: 0.2 ms of a 10 ms step)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (dev: gloo)")
    ap.add_argument("--same-gpu", action="store_true", help="dev only: all ranks on cuda:0")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="dev only: with --gpus N > 1 keep going on the Python driver if the native RCCL loop is unavailable")
    ap.add_argument("--samples-by", choices=("bcast", "local"), default="bcast",
                    help="--gpus N > 1: X reaches the ranks by the library's RCCL broadcast (default) or every rank uploads its own copy")
    ap.add_argument("--launch-check", action="store_true",
                    help="dev / CPU test of the self-launch: the ranks rendezvous over gloo, rank 0 prints one line, no GPU touched")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (torch.distributed.run as a CHILD process,
    # before this process has imported torch or touched a GPU), relay rank 0's single JSON line and the children's status
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))
    if args.launch_check:
        raise SystemExit(launch_check(args.gpus))

    # stdout carries exactly ONE line, the result: libraries that chat on fd 1 (RCCL prints a version
    # banner when a communicator is created, gloo its peer counts) are sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    import chbin_amd  # noqa: F401
    from chbin_amd import _lib, synth
    from chbin_amd import distributed as cdist_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    use_dist = world > 1
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    xdev = dev if args.backend == "nccl" else None      # where exchanged tensors live

    N, D, B, m = args.contigs, args.dim, args.bins, args.neighbors
    S = coverage_columns(D)
    X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0, mix=args.mix, sigma=args.sigma)
    perms = synth.draw_permutations(initial, 10, seed=0)     # np.random.seed(0): ch_bin.py:22
    n_move = perms.shape[1]
    qp_per_step = n_move * B

    ctx = _lib.Context(local_rank)

    # N > 1: contigs of every batch sharded across the ranks inside the C++ loop, label slices
    # exchanged with RCCL all-gathers.  The Python driver (torch.distributed all_reduce between
    # rounds) exists for development only: a driver run that falls back to it FAILS.
    native = False
    if use_dist:
        err = None
        try:
            cdist_mod.init_native_comm(ctx, device=xdev)
            native = True
        except Exception as e:  # noqa: BLE001
            err = e
        flag = torch.tensor([1 if native else 0], device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native = bool(flag.item())
        if not native:
            if rank == 0:
                print(f"native RCCL communicator unavailable ({err})", file=sys.stderr)
            if not args.allow_fallback:
                dist.barrier()
                dist.destroy_process_group()
                raise SystemExit(3)
    samples_by = "local"
    if native and args.samples_by == "bcast":
        # the feature matrix crosses the host boundary ONCE (rank 0) and reaches the other GPUs by the
        # library's RCCL broadcast over xGMI (chb_bcast_samples); every rank then keeps its own resident copy.  The
        # library makes the ranks agree on {status, N, D, root} before the collective, so a failure is an error on EVERY
        # rank -- and then each rank uploads its own (identical, synthetic) copy instead, and the line says so
        try:
            ctx.bcast_samples(X if rank == 0 else None, N, D, root=0)
            samples_by = "bcast"
        except _lib.ChbError as e:
            print(f"rank {rank}: chb_bcast_samples failed ({e}); every rank uploads its own copy", file=sys.stderr)
            ctx.set_samples(X)
    else:
        ctx.set_samples(X)                                    # resident in HBM before timing

    def one_step():
        if use_dist and not native:
            return cdist_mod.fit_cluster_distributed(ctx, X, B, initial, perms[:1], m, 1,
                                                     batch=args.batch, device=xdev)[0]
        return ctx.fit_cluster(B, initial, perms[:1], m, 1, batch=args.batch)[0]

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        labels1 = one_step()
    ctx.profile_reset()
    # (the event pairs around the two dominant kernels -- the roofline's live launch durations -- cost the stream about
    #  8 us per launch, 0.2 ms of a 10 ms step: they are recorded in the LAST timed step only, i.e. around every launch
    #  of one whole sweep; --kernel-events-every-step keeps them on throughout)
    event_steps = 0 if args.no_kernel_events else (args.steps if args.kernel_events_every_step else min(1, args.steps))
    ctx.profile_enable(2 if event_steps == args.steps and event_steps > 0 else 0)    # events around the two dominant kernels only
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if 0 < event_steps < args.steps and i == args.steps - event_steps:
            ctx.profile_enable(2)
        labels1 = one_step()
    sync()
    dt = time.perf_counter() - t0
    ctx.profile_enable(0)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / max(args.steps, 1) * 1e3
    try:
        batch_used = int(ctx.counter("batch_size"))
    except Exception:  # noqa: BLE001  (a library from before the counter: CHBIN_LIB A/B runs)
        batch_used = args.batch or 8192
    value = qp_per_step * args.steps / dt
    dom = {k: ctx.profile_get(k) for k in ("prefilter", "hull_qp")}     # measured INSIDE the timed region
    stats = ctx.fit_stats()
    # evidence of the exchange the timed steps ran over: what RCCL itself reports for the communicator, and how the
    # hull evaluations of the last timed step were spread over the ranks (disjoint slices: they add up)
    evidence = None
    if use_dist:
        mine = {"rank": rank, "comm": ctx.comm_info(), "hull_evaluated_last_step": int(stats["hull_evaluated"]),
                "hull_needed_last_step": int(stats["hull_needed"]), "rounds_last_step": int(stats["rounds"])}
        allv = [None] * world
        dist.all_gather_object(allv, mine)
        evidence = {"transport": allv[0]["comm"]["transport"],
                    "comm_ranks_reported_by_rccl": [v["comm"]["comm_ranks"] for v in allv],
                    "hull_evaluated_per_rank_last_step": [v["hull_evaluated_last_step"] for v in allv],
                    "hull_needed_last_step": allv[0]["hull_needed_last_step"],
                    "rounds_per_rank_last_step": [v["rounds_last_step"] for v in allv]}

    # ---- untimed pass with an event pair around every kernel: the full table
    names = ("prefilter", "rescore", "prefilter_update", "rescore_update", "query_norms", "fit_start", "topm_fallback", "topm_base",
             "topm_update", "hull_qp", "slow_path", "argmin", "bucket", "pool", "prefilter_retry")
    prof_steps = max(1, min(args.steps, 3))
    ctx.profile_reset()
    ctx.profile_enable(1)
    for _ in range(prof_steps):
        one_step()
    sync()
    ctx.profile_enable(0)
    prof = {k: ctx.profile_get(k) for k in names}

    out = None
    if rank == 0:
        # shadow row width of the shortlist stage: 144 / 160 columns, or two to four 144-column slices (wide rows, D <= 573)
        Dz = 144 if D + 3 <= 144 else (160 if D + 3 <= 160 else 144 * ((D + 3 + 143) // 144))
        fused = ctx.counter("fused_enabled") == 1
        bytes_qp = 8.0 * (m * D + D / B + 1)      # no-reuse gather model per hull distance
        x_mb = N * ((D + 15) // 16 * 16) * 8 / 1e6      # (rows of the resident matrix are padded to whole 128-byte lines)
        gather_peak = gather_ceiling_gbs(x_mb)
        gather_note = ("achieved = ALGORITHMIC bytes of SURVEY 8(d)'s no-reuse gather model (every candidate row counted as "
                       f"read from memory) / the measured launch time, peak = the 8 TB/s HBM spec.  The rows come out of a {x_mb:.0f} MB "
                       "resident matrix, i.e. from the XCDs' L2 and the 256 MiB Infinity Cache rather than HBM: `traffic` is what "
                       "really crossed the fabric, and `gather_ceiling` is MI355X_MICROARCH.md's indexed-row gather rate for a "
                       "table of this size")

        def gather_entry(name, p, bytes_unit, unit_name, note):
            ach = p["work"] * bytes_unit / (p["ms"] * 1e-3) / 1e9
            return {"kernel": name, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ach / HBM_PEAK_GBS, "traffic": None,
                    "gather_ceiling": {"peak": gather_peak, "unit": "GB/s", "frac": ach / gather_peak,
                                       "table_mb": x_mb},
                    "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"], "total_ms": p["ms"],
                    unit_name + "_per_s_kernel": p["work"] / (p["ms"] * 1e-3), "bytes_per_" + unit_name: bytes_unit,
                    "algorithmic_bytes_per_launch": p["work"] * bytes_unit / p["launches"],
                    "note": note}

        def kernel_entries(pr):
            kern = []
            for name in ("topm_base", "topm_update"):
                p = pr.get(name)
                if p and p["launches"]:
                    ach = p["work"] * 3.0 * ((D + 7) // 8 * 8) / (p["ms"] * 1e-3) / 1e12
                    kern.append({"kernel": name, "bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                                 "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                                 "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                                 "total_ms": p["ms"],
                                 "note": "brute-force selection: fp64 VALU, non-fused sub/mul/add (1 flop per instruction)"})
            p = pr.get("prefilter")
            if p and p["launches"]:
                # fp16 MFMA dot products, 2*Dz flops per (query, member) pair counted ONCE (the kernel streams
                # a bin's members twice -- threshold sweep, then shortlist sweep)
                ach = p["work"] * 2.0 * Dz / (p["ms"] * 1e-3) / 1e12
                kern.append({"kernel": "prefilter", "bound": "mfma", "achieved": ach, "peak": F16_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": ach / F16_PEAK_TFLOPS, "traffic": None,
                             "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                             "total_ms": p["ms"], "pairs_per_s": p["work"] / (p["ms"] * 1e-3),
                             "note": "fp16 v_mfma_f32_32x32x16 shortlist; waits on the member-tile LDS-DMA stream and the "
                                     "per-tile barrier, not the matrix core, set its time (profiles/README.md)"})
            p = pr.get("hull_qp")
            if p and p["launches"]:
                note = ("fused selection + hull distance: every shortlisted row is gathered once; " if fused else
                        "hull distance on the selected lists; ") + gather_note
                kern.append(gather_entry("hull_qp", p, bytes_qp, "qp", note))
            for name in ("rescore", "rescore_update"):
                p = pr.get(name)
                if p and p["launches"]:
                    kern.append(gather_entry(name, p, 8.0 * m * D, "pair",
                                             "exact cdist-rounded distances on the shortlists (at least the m winners' "
                                             "rows have to be read); " + gather_note))
            for name in ("prefilter_update", "query_norms", "fit_start", "topm_fallback", "slow_path", "argmin", "bucket", "pool", "prefilter_retry"):
                p = pr.get(name)
                if p and p["launches"]:
                    kern.append({"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS,
                                 "unit": "GB/s", "frac": None, "traffic": None,
                                 "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                                 "total_ms": p["ms"]})
            return kern

        kern = kernel_entries(prof)
        for k in kern:   # per step, so that the two passes are comparable
            k["ms_per_step"] = k["total_ms"] / prof_steps
        # HBM-side traffic per launch from the committed PMC profile of this round (separate rocprofv3 --pmc
        # passes, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null when unavailable
        # (quoted only while the kernel sources are byte-identical to the build the counters were collected on, and
        #  only for the configuration they were collected on; otherwise null)
        traffic = {}
        onchip = {}
        traffic_note = None
        try:
            default_tf = TRAFFIC_FILE if m <= 5 else TRAFFIC_FILE.replace("_traffic", "_m15_traffic")
            cfg_idx = next((i for i, shp in BASELINE_CONFIGS.items() if shp == (N, D, B)), None)
            if cfg_idx in (3, 4) and m <= 5:
                default_tf = TRAFFIC_FILE.replace("_traffic", f"_cfg{cfg_idx}_traffic")
            wide528 = (N, D, B) == (100_000, 528, 64) and m == 5   # the wide-row run of the profile set (KmerK = 5 width)
            if wide528:
                default_tf = TRAFFIC_FILE.replace("_traffic", "_wide528_traffic")
            tj = json.load(open(args.traffic_file or os.path.join(ROOT, "profiles", default_tf)))
            tf_label = (f"{args.traffic_file} (this run's own rocprofv3 --pmc passes; committed as profiles/{default_tf})"
                        if args.traffic_file else f"profiles/{default_tf}")
            tr = tj["kernels"]
            # (kernel names as rocprofv3 prints them: the shortlist kernel's template list has grown over the rounds)
            ml = 5 if m <= 5 else (8 if m <= 8 else 16)
            # (the base shortlist launch of this configuration: the instantiation with the most launches in the table among
            #  the ordinary base builds -- tile skipping and threshold pools are chosen per fit)
            base_sl = sorted((k for k in tr if k.startswith(f"shortlist_kernel<{ml}, false, ") and ", 0, " in k and not k.endswith("true>")),
                             key=lambda k: -tr[k].get("launches", 0))
            if Dz > 160:   # the wide builds: shortlist_wide_kernel<ML, UPD, slices>
                base_sl = [f"shortlist_wide_kernel<{ml}, false, {Dz // 144}>"]
            tmap = {"prefilter": base_sl + [f"shortlist_kernel<{ml}, false, 9, 0, false>", f"shortlist_kernel<{ml}, false, 9, 0>",
                                            f"shortlist_kernel<{ml}, false, 9>"],
                    "hull_qp": ["hull_select_qp_kernel<5, 7, 4, true>"] if m <= 5 else
                               ["hull_select_qp16_kernel<4, false>", "hull_select_qp16_kernel<4>"],
                    "prefilter_update": ([f"shortlist_wide_kernel<{1 if m <= 8 else 2}, true, {Dz // 144}>"] if Dz > 160 else []) +
                                        [f"shortlist_kernel<{1 if m <= 8 else 2}, true, {9 if D <= 141 else 10}, 0, false, false, false>",
                                         f"shortlist_kernel<{1 if m <= 8 else 2}, true, 9, 0, false, false>",
                                         f"shortlist_kernel<{1 if m <= 8 else 2}, true, 9, 0, false>",
                                         "shortlist_kernel<1, true, 9, 0>", "shortlist_kernel<1, true, 9>"]}
            stamp = kernel_source_stamp()
            if tj.get("kernel_source_stamp") != stamp:
                traffic_note = (f"{tf_label} was measured on kernel sources {tj.get('kernel_source_stamp')}, "
                                f"this build is {stamp}: not quoted")
            elif (((N, D, B) == (100_000, 136, 64) and m in (5, 15)) or (cfg_idx in (3, 4) and m == 5) or wide528) and not args.batch and not use_dist and fused:
                for name, names in tmap.items():
                    src = next((k for k in names if k in tr), None)
                    if src is not None:
                        label = f"{tf_label} ({src}; kernel sources {stamp})"
                        if tr[src].get("traffic_bytes_per_launch") is not None:
                            traffic[name] = (tr[src]["traffic_bytes_per_launch"], label)
                        onchip[name] = (tr[src], label)
        except Exception as e:  # noqa: BLE001
            traffic_note = f"no traffic file ({e})"
        for k in kern:
            if k["kernel"] in traffic:
                k["traffic"], k["traffic_source"] = traffic[k["kernel"]]

        def limiter_of(entry):
            """What the kernel is really bound by, from counters of THIS build (the contract's `frac` prices SURVEY 8(d)'s
            no-reuse byte model against the HBM peak and can exceed 1 when the rows come out of L2 / Infinity Cache): every
            measured resource with its own peak; `resource` names the busiest one."""
            oc = onchip.get(entry["kernel"])
            if oc is None:
                return None
            t, label = oc
            secs = entry["avg_launch_ms"] * 1e-3
            cands = []
            if t.get("l2_read_bytes_per_launch"):
                ach = t["l2_read_bytes_per_launch"] / secs / 1e9
                c = {"resource": "L2 -> CU reads (row gather / tile DMA: vector-L1 read requests x 128 B / launch time)", "achieved": ach,
                     "peak": L2_GATHER_PEAK_GBS, "unit": "GB/s", "frac": ach / L2_GATHER_PEAK_GBS,
                     "peak_source": "MI355X_MICROARCH.md: 16.8-18.8 TB/s for indexed rows of an L2-resident table",
                     "frac_of_l2_aggregate_34.5TBs": ach / L2_AGGREGATE_GBS,
                     "l2_read_bytes_per_launch": t["l2_read_bytes_per_launch"], "l2_hit_rate": t.get("l2_hit_rate")}
                if entry.get("algorithmic_bytes_per_launch"):
                    c["requested_over_algorithmic"] = t["l2_read_bytes_per_launch"] / entry["algorithmic_bytes_per_launch"]
                cands.append(c)
            if t.get("traffic_bytes_per_launch"):
                ach = t["traffic_bytes_per_launch"] / secs / 1e9
                cands.append({"resource": "HBM / Infinity Cache side (2 x FETCH_SIZE + WRITE_SIZE)", "achieved": ach,
                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS})
            if t.get("valu_busy") is not None:
                cands.append({"resource": "vector ALU issue (SQ_ACTIVE_INST_VALU / SQ_BUSY_CU_CYCLES)", "achieved": t["valu_busy"],
                              "peak": 1.0, "unit": "share of SIMD time", "frac": t["valu_busy"]})
            if t.get("mfma_busy"):
                cands.append({"resource": "matrix pipe (SQ_VALU_MFMA_BUSY_CYCLES / 4 SQ_BUSY_CU_CYCLES)", "achieved": t["mfma_busy"],
                              "peak": 1.0, "unit": "share of SIMD time", "frac": t["mfma_busy"],
                              "coexec_share_of_mfma_busy": t.get("mfma_coexec")})
            if not cands:
                return None
            top = dict(max(cands, key=lambda c: c["frac"]))
            top["source"] = label + "; duration: HIP events of this run"
            top["wave_time_shares"] = {k2: t.get(k2) for k2 in ("wait_any", "wait_inst", "active")}
            top["candidates"] = [{k2: c[k2] for k2 in ("resource", "achieved", "peak", "unit", "frac")} for c in cands]
            return top

        kern.sort(key=lambda k: -k["total_ms"])

        # the roofline object: the dominant kernel, with the durations measured inside the timed region
        roofline = None
        timed = kernel_entries(dom)
        if timed:
            timed.sort(key=lambda k: -k["total_ms"])
            roofline = dict(timed[0])
            roofline["ms_per_step"] = roofline["total_ms"] / max(event_steps, 1)
            roofline["measured"] = ("HIP events on the library's stream around every launch of the kernel in the last %d of "
                                    "the %d timed steps" % (event_steps, args.steps))
            if roofline["kernel"] in traffic:
                roofline["traffic"], roofline["traffic_source"] = traffic[roofline["kernel"]]
            elif traffic_note:
                roofline["traffic_note"] = traffic_note
            roofline["limiter"] = limiter_of(roofline)
            if len(timed) > 1:
                o = timed[1]
                roofline["runner_up"] = {"kernel": o["kernel"], "ms_per_step": o["total_ms"] / max(event_steps, 1),
                                         "frac": o["frac"], "bound": o["bound"], "unit": o["unit"],
                                         "achieved": o["achieved"], "peak": o["peak"], "limiter": limiter_of(o)}
        # whole-path view in SURVEY 8(d)'s no-reuse gather model against the HBM spec: every hull distance the
        # sweep needs x bytes_QP over the sweep's wall time (the north star's 40 % target is on this figure)
        path_roofline = {"bound": "hbm", "achieved": value * bytes_qp / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": value * bytes_qp / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_qp": bytes_qp,
                         # BASELINE.md 3 / SURVEY 8(d): the same figure against the measured-copy peak of the part
                         "measured_copy_peak": HBM_COPY_PEAK_GBS,
                         "frac_of_measured_copy_peak": value * bytes_qp / 1e9 / HBM_COPY_PEAK_GBS}

        # ---- end-to-end bin-assign wall clock (all sweeps until no label changes, max 10)
        e2e = None
        if not args.no_e2e and not use_dist:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            lab_full, its, changed = ctx.fit_cluster(B, initial, perms, m, 10, batch=args.batch)
            torch.cuda.synchronize()
            e2e = {"seconds": time.perf_counter() - t1, "sweeps": int(its),
                   "changed_per_sweep": [int(c) for c in changed],
                   "accuracy_vs_truth": float((lab_full == true).mean())}

        # ---- CPU baseline: the oracle (scalar C restatement of the reference path) on the first
        # contigs of the same sweep, full N and all bins, one host thread.
        cpu = None
        if not use_dist and args.cpu_sample > 0:
            from oracle import oracle as O
            ns = min(args.cpu_sample, n_move)
            t2 = time.perf_counter()
            lab_o, _ = O.sweep(X, B, initial, perms[0][:ns], m)
            cdt = time.perf_counter() - t2
            ids = perms[0][:ns]
            if not np.array_equal(lab_o[ids], labels1[ids]):
                raise SystemExit("PARITY FAILURE: GPU labels differ from the oracle on the sampled prefix")
            cpu = {"value": ns * B / cdt, "unit": "QP/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} contigs of the same sweep-1 permutation x all {B} bins "
                             f"against the full N={N} (oracle/chb_oracle.c:chbo_sweep, {cdt:.1f} s); "
                             "labels of the sample verified identical to the GPU's",
                   "host_cpus": os.cpu_count()}
            # "at run time probe `import quadprog`; only if present also time genuine quadprog" -- and
            # always quote the reference's own non-solver Python overhead beside the port's figure
            try:
                import quadprog  # noqa: F401
                cpu["quadprog"] = "importable (not timed: the port above is the baseline this line reports)"
            except Exception as e:  # noqa: BLE001
                cpu["quadprog"] = f"unavailable ({type(e).__name__}): the reference's solver cannot be timed on this box"
            cpu["reference_python_overhead_us_per_qp"] = 195
            cpu["reference_python_overhead_source"] = ("SURVEY.md 8(a)/8(d): 124 us find_nearest_from_cluster + 71 us "
                                                       "hull_distance glue per (contig, bin), measured with a no-op solver "
                                                       "=> <= 5.1e3 QP/s per core for the reference, before quadprog")
            # same arithmetic on all the host cores this process may use: the sampled
            # contigs are evaluated independently against the frozen seed-state labels, so this times
            # the per-(contig, bin) work of a sweep in parallel; it is not itself a sequential sweep
            try:
                affinity = len(os.sched_getaffinity(0))
            except AttributeError:
                affinity = os.cpu_count() or 1
            nthr = max(1, min(affinity, 16))   # a one-GPU box's CPU share is 16 cores, whatever the host has
            cpu["affinity_cpus"] = affinity
            # parity of the TIMED configuration beyond the prefix: the end-to-end fit above ran exactly like the timed
            # steps (look-ahead across batches on); once it stopped at a sweep that changed nothing, every movable
            # contig's label must be the strict-'>' argmin of its hull distances given everyone else's final label
            # (algorithm.py:46-60) -- checked by the oracle on a random sample of the final labels
            if e2e is not None and e2e["changed_per_sweep"] and e2e["changed_per_sweep"][-1] == 0:
                rng = np.random.default_rng(7)
                fp_ids = rng.choice(np.flatnonzero(initial < 0), min(384, n_move), replace=False)
                bb_o, _ = O.eval_frozen_mt(X, B, lab_full, fp_ids, m, nthr)
                if not np.array_equal(bb_o, lab_full[fp_ids]):
                    raise SystemExit("PARITY FAILURE: final labels of the end-to-end fit are not a fixed point of the "
                                     "reference sweep on the oracle's sample")
                cpu["fixed_point_check"] = (f"{len(fp_ids)} random movable contigs of the converged end-to-end fit: label == "
                                            "oracle argmin over all bins given the other final labels")
            if nthr > 1:
                ids_mt = perms[0][:min(ns * nthr, n_move)]
                t3 = time.perf_counter()
                O.eval_frozen_mt(X, B, initial, ids_mt, m, nthr)
                mdt = time.perf_counter() - t3
                cpu["multi_core"] = {"value": len(ids_mt) * B / mdt, "unit": "QP/s", "cores": nthr, "kind": "port",
                                     "cores_note": f"{nthr} threads = this box's CPU share (affinity {affinity}, host {os.cpu_count()})",
                                    "sample": f"{len(ids_mt)} contigs x all {B} bins against the full N={N}, frozen "
                                              f"seed-state labels, OpenMP over contigs (chbo_eval_frozen_mt, {mdt:.1f} s)"}

        # ---- extra legs (never part of `value`): the reference's function-default neighbour count, bins that
        # overlap (speculation has to repeat rounds), and how far the winning bin is ahead of the runner-up
        extra = None
        if not args.no_extra and not use_dist:
            extra = {}

            def sweep_time(c, init_l, perms_l, mm, reps=2):
                c.fit_cluster(B, init_l, perms_l[:1], mm, 1)
                torch.cuda.synchronize()
                ta = time.perf_counter()
                for _ in range(reps):
                    c.fit_cluster(B, init_l, perms_l[:1], mm, 1)
                torch.cuda.synchronize()
                return (time.perf_counter() - ta) / reps

            # (a) AlgoNumNeighbors = 15 (algorithm.py:17, cli/clustering.py:23)
            if m != 15:
                t15 = sweep_time(ctx, initial, perms, 15)
                b15 = 8.0 * (15 * D + D / B + 1)
                extra["neighbors_15"] = {"ms_per_sweep": t15 * 1e3, "qp_per_s": qp_per_step / t15,
                                         "path_frac_of_hbm_peak": qp_per_step / t15 * b15 / 1e9 / HBM_PEAK_GBS,
                                         "shortlist_overflows": ctx.counter("prefilter_overflow")}
            # (b) label margins on the benchmark data: second-best minus best hull distance at every movable
            # contig's last visit of the whole fit.  The GPU solver agrees with the oracle's Goldfarb-Idnani to
            # ~1e-12; a label could only differ from real quadprog's where the margin is within solver rounding.
            if hasattr(ctx, "fit_cluster_margins"):
                lab_m, its_m, margin = ctx.fit_cluster_margins(B, initial, perms, m, 10, batch=args.batch)
                mv = margin[initial < 0]
                mv = mv[np.isfinite(mv)]
                extra["label_margin"] = {"visits": int(mv.size), "sweeps": int(its_m), "min": float(mv.min()),
                                         "median": float(np.median(mv)),
                                         "below_1e-9": int((mv < 1e-9).sum()), "below_1e-7": int((mv < 1e-7).sum()),
                                         "below_1e-5": int((mv < 1e-5).sum()),
                                         "note": "d2 - d1 (runner-up minus winning hull distance) at each movable contig's "
                                                 "last visit; QP distances agree with the CPU oracle to <= 1e-9 (tests), "
                                                 "north-star tolerance 1e-5"}
            # (c) overlapping bins.  Two generator settings: one where the bins overlap but the algorithm still
            # recovers them (speculation has to repeat rounds), and one where the reference algorithm itself
            # collapses into a few giant bins (bin-size skew on top of failing speculation).
            for key, mixh, sigh in (("overlapping_bins", 0.3, 4.5e-3), ("collapsed_bins", 0.5, 6e-3)):
                Xh, inith, trueh = synth.make_synthetic(N, D, B, S=S, seed=0, mix=mixh, sigma=sigh)
                permsh = synth.draw_permutations(inith, 3, seed=0)
                ctx.set_samples(Xh)
                th = sweep_time(ctx, inith, permsh, m)
                sth = ctx.fit_stats()
                labh, _, _ = ctx.fit_cluster(B, inith, permsh[:1], m, 1)
                sizes = np.sort(np.bincount(labh[labh >= 0], minlength=B))[::-1]
                extra[key] = {"generator": f"mix={mixh} sigma={sigh}", "ms_per_sweep": th * 1e3,
                              "qp_per_s": int(permsh.shape[1]) * B / th,
                              "rounds_per_batch": sth["rounds"] / max(sth["batches"], 1),
                              "hull_evaluated_per_needed": sth["hull_evaluated"] / max(sth["hull_needed"], 1),
                              "accuracy_vs_truth_after_sweep_1": float((labh == trueh).mean()),
                              "largest_bins": [int(v) for v in sizes[:3]]}

        out = {
            "metric": f"convex-hull QP distances/sec (+ end-to-end bin-assign wall-clock), N={N} D={D} B={B}",
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{config_name(N, D, B)}: synthetic {N} contigs x D={D} x {B} bins, "
                                   f"AlgoNumNeighbors={m}, one full fit_cluster sweep from the seed "
                                   "state per step (exact sequential label semantics)",
                       "n_contigs": N, "dim": D, "bins": B, "neighbors": m, "movable": int(n_move),
                       "qp_per_step": int(qp_per_step), "batch": batch_used,
                       "generator": {"mix": args.mix, "sigma": args.sigma, "coverage_columns": S},
                       "parallelism_evidence": evidence,
                       "parallelism": (f"contig-sharded x{world}, " + ("RCCL all-gather in the C++ loop, X " + ("by the library's RCCL broadcast (chb_bcast_samples)" if samples_by == "bcast" else "uploaded by every rank")
                                                                       if native else "torch.distributed all_reduce (Python driver)"))
                       if use_dist else "single GPU"},
            "roofline": roofline,
            "path_roofline": path_roofline,
            "kernels": kern,
            "kernels_source": f"separate untimed pass of {prof_steps} step(s) with an event pair around every kernel",
            "cpu_baseline": cpu,
            "end_to_end_bin_assign": e2e,
            "fit_stats_last_call": stats,
            "prefilter": {"enabled": ctx.counter("prefilter_enabled"), "fused": int(fused),
                          "shortlist_overflows_last_call": ctx.counter("prefilter_overflow")},
            "extra": extra,
        }
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
