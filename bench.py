#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: convex-hull QP distances/sec (+ end-to-end bin-assign
wall-clock) on the synthetic N=100k x D=136 x B=64 workload (BASELINE.json configs[2]).

A "step" is ONE complete fit_cluster sweep (algorithm.py:43-60) from the seed state: every movable
contig (~98k) is visited in the reference's permutation order and, for each of the 64 bins, its m
nearest members are selected and the point-to-convex-hull QP distance is evaluated
(~6.27M hull distances), with the reference's sequential label semantics.  The feature matrix is
resident in HBM before the timed region; labels/permutation cross the boundary every step exactly
as the reference's call does.  `value` = hull distances the sequential loop needs / wall time
(speculative re-evaluations are NOT counted as work).

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

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector == matrix peak (AMD public spec; the local guide lists no fp64 row)
F16_PEAK_TFLOPS = 2500.0  # dense fp16/bf16 MFMA peak (MI355X_MICROARCH.md)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


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
    ap.add_argument("--cpu-sample", type=int, default=300)
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="dev: no per-kernel HIP events in the timed steps")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (dev: gloo)")
    ap.add_argument("--same-gpu", action="store_true", help="dev only: all ranks on cuda:0")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the result: libraries that chat on fd 1 (RCCL prints a version
    # banner when a communicator is created, gloo its peer counts) are sent to stderr instead
    sys.stdout.flush()
    result_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch

    import chbin_amd
    from chbin_amd import _lib, synth
    from chbin_amd import distributed as cdist_mod

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_gpu:
        local_rank = 0
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
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
    S = 1 if D <= 136 else (5 if D == 140 else 10)
    X, initial, true = synth.make_synthetic(N, D, B, S=S, seed=0)
    perms = synth.draw_permutations(initial, 10, seed=0)     # np.random.seed(0): ch_bin.py:22
    n_move = perms.shape[1]
    qp_per_step = n_move * B

    ctx = _lib.Context(local_rank)
    ctx.set_samples(X)                                        # resident in HBM before timing

    # N > 1: contigs of every batch sharded across the ranks inside the C++ loop, label slices
    # exchanged with RCCL all-gathers; if the native communicator cannot be created the Python
    # driver (torch.distributed all_reduce between rounds) is used instead.
    native = False
    if use_dist:
        try:
            cdist_mod.init_native_comm(ctx, device=xdev)
            native = True
        except Exception as e:  # noqa: BLE001
            if rank == 0:
                print(f"native RCCL communicator unavailable ({e}); using the Python driver", file=sys.stderr)
        flag = torch.tensor([1 if native else 0], device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        native = bool(flag.item())

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
    ctx.profile_enable(not args.no_kernel_events)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        labels1 = one_step()
    sync()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    if use_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / max(args.steps, 1) * 1e3
    value = qp_per_step * args.steps / dt

    prof = {k: ctx.profile_get(k) for k in ("prefilter", "rescore", "prefilter_update", "rescore_update", "query_norms",
                                            "topm_fallback", "topm_base", "topm_update", "hull_qp", "slow_path", "argmin",
                                            "bucket")}
    stats = ctx.fit_stats()

    out = None
    if rank == 0:
        Dp = (D + 7) // 8 * 8
        kern = []
        # distance/top-m tiles: 3 fp64 flops (sub, mul, add -- deliberately unfused to round like
        # cdist) per (query, member, feature); work unit recorded per launch = (query, member) pairs
        for name in ("topm_base", "topm_update"):
            p = prof[name]
            if p["launches"]:
                flops = p["work"] * 3.0 * Dp
                ach = flops / (p["ms"] * 1e-3) / 1e12
                kern.append({"kernel": name, "bound": "mfma", "achieved": ach, "peak": FP64_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": ach / FP64_PEAK_TFLOPS, "traffic": None,
                             "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                             "total_ms": p["ms"],
                             "note": "fp64 VALU (non-fused sub/mul/add, 1 flop per instruction); "
                                     "peak is the fp64 vector==matrix FMA peak, so 0.5 is the ceiling"})
        # shortlist stage: fp16 MFMA dot products, 2*Dz flops per (query, member) pair (one pass; the
        # kernel streams a bin's members twice -- threshold sweep, then shortlist sweep)
        p = prof["prefilter"]
        if p["launches"]:
            Dz = 144 if D <= 144 else 160
            ach = p["work"] * 2.0 * Dz / (p["ms"] * 1e-3) / 1e12
            kern.append({"kernel": "prefilter", "bound": "mfma", "achieved": ach, "peak": F16_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": ach / F16_PEAK_TFLOPS, "traffic": None,
                         "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                         "total_ms": p["ms"], "pairs_per_s": p["work"] / (p["ms"] * 1e-3),
                         "note": "fp16 v_mfma_f32_32x32x16 shortlist; LDS fragment reads and the "
                                 "selection VALU work, not the matrix core, set its time"})
        # exact rescoring of the shortlists: at least the m winners' rows have to be read to know their
        # exact distances, so the algorithmic bytes per (position, bin) pair are m * D * 8 (the
        # shortlist itself is ~5.2 rows at m = 5); same no-reuse gather model as the hull QP
        p = prof["rescore"]
        if p["launches"]:
            bytes_pair = 8.0 * m * D
            ach = p["work"] * bytes_pair / (p["ms"] * 1e-3) / 1e9
            kern.append({"kernel": "rescore", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                         "total_ms": p["ms"], "pairs_per_s": p["work"] / (p["ms"] * 1e-3),
                         "bytes_per_pair": bytes_pair,
                         "note": "gather of the shortlisted rows (fp64, 8*D bytes each), unfused sequential "
                                 "sums; no-reuse model, rows come largely out of L2 / Infinity Cache"})
        for name in ("prefilter_update", "rescore_update", "query_norms", "topm_fallback", "slow_path", "argmin", "bucket"):
            p = prof[name]
            if p["launches"]:
                kern.append({"kernel": name, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS,
                             "unit": "GB/s", "frac": None, "traffic": None,
                             "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                             "total_ms": p["ms"]})
        p = prof["hull_qp"]
        if p["launches"]:
            bytes_qp = 8.0 * (m * D + D / B + 1)
            ach = p["work"] * bytes_qp / (p["ms"] * 1e-3) / 1e9
            kern.append({"kernel": "hull_qp", "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                         "avg_launch_ms": p["ms"] / p["launches"], "launches": p["launches"],
                         "total_ms": p["ms"], "qp_per_s_kernel": p["work"] / (p["ms"] * 1e-3),
                         "bytes_per_qp": bytes_qp,
                         "note": "achieved = SURVEY 8(d)'s no-reuse gather model (bytes_QP per hull distance); "
                                 "vertex rows are re-used out of L2 / Infinity Cache, so it can exceed the HBM "
                                 "peak -- `traffic` is what actually crossed the fabric per launch"})
        # HBM-side traffic per launch from the committed PMC profile (separate rocprofv3 --pmc passes,
        # FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950); null when unavailable
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01e_traffic.json")))["kernels"]
            tmap = {"prefilter": "shortlist_kernel<5, false, 9>", "hull_qp": "hull_qp_kernel<5, 4, false>",
                    "rescore": "rescore_kernel<8, 2>", "rescore_update": "rescore_kernel<8, 2>",
                    "prefilter_update": "shortlist_kernel<1, true, 9>",
                    "query_norms": "query_norms_kernel"}
            if (N, D, B, m) == (100_000, 136, 64, 5) and (args.batch or 8192) == 8192 and not use_dist:
                for k in kern:
                    src = tmap.get(k["kernel"])
                    if src in tr:
                        k["traffic"] = tr[src]["traffic_bytes_per_launch"]
                        k["traffic_source"] = "profiles/r01e_traffic.json (" + src + ")"
        except Exception:  # noqa: BLE001
            pass
        kern.sort(key=lambda k: -k["total_ms"])
        roofline = dict(kern[0]) if kern else None
        # whole-path view in SURVEY 8(d)'s no-reuse gather model: every hull distance the sweep needs
        # x bytes_QP over the sweep's wall time (selection traffic is on top of that and is not part
        # of bytes_QP because no distance row is ever materialised)
        bytes_qp = 8.0 * (m * D + D / B + 1)
        path_roofline = {"bound": "hbm", "achieved": value * bytes_qp / 1e9, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": value * bytes_qp / 1e9 / HBM_PEAK_GBS,
                         "bytes_per_qp": bytes_qp}

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
            # same arithmetic on all the host cores this process may use: the sampled
            # contigs are evaluated independently against the frozen seed-state labels, so this times
            # the per-(contig, bin) work of a sweep in parallel; it is not itself a sequential sweep
            try:
                nthr = max(1, min(len(os.sched_getaffinity(0)), 16))   # the GPU box grants ~16 cores per GPU
            except AttributeError:
                nthr = max(1, min(os.cpu_count() or 1, 16))
            if nthr > 1:
                ids_mt = perms[0][:min(ns * nthr, n_move)]
                t3 = time.perf_counter()
                bb_mt, _ = O.eval_frozen_mt(X, B, initial, ids_mt, m, nthr)
                mdt = time.perf_counter() - t3
                cpu["all_cores"] = {"value": len(ids_mt) * B / mdt, "unit": "QP/s", "cores": nthr, "kind": "port",
                                    "sample": f"{len(ids_mt)} contigs x all {B} bins against the full N={N}, frozen "
                                              f"seed-state labels, OpenMP over contigs (chbo_eval_frozen_mt, {mdt:.1f} s)"}

        out = {
            "metric": "convex-hull QP distances/sec (+ end-to-end bin-assign wall-clock), N=100k D=136 B=64",
            "value": value, "unit": "QP/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"BASELINE configs[2]: synthetic {N} contigs x D={D} x {B} bins, "
                                   f"AlgoNumNeighbors={m}, one full fit_cluster sweep from the seed "
                                   "state per step (exact sequential label semantics)",
                       "n_contigs": N, "dim": D, "bins": B, "neighbors": m, "movable": int(n_move),
                       "qp_per_step": int(qp_per_step), "batch": args.batch or 8192,
                       "parallelism": (f"contig-sharded x{world}, " + ("RCCL all-gather in the C++ loop" if native else "torch.distributed all_reduce")) if use_dist else "single GPU"},
            "roofline": roofline,
            "path_roofline": path_roofline,
            "kernels": kern,
            "cpu_baseline": cpu,
            "end_to_end_bin_assign": e2e,
            "fit_stats_last_call": stats,
            "prefilter": {"enabled": ctx.counter("prefilter_enabled"),
                          "shortlist_overflows_last_call": ctx.counter("prefilter_overflow")},
        }
        result_out.write(json.dumps(out) + "\n")
        result_out.flush()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
