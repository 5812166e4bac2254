#!/usr/bin/env python
"""bench.py -- BO-iterations/sec (fit + argmax) for the 16-16-1 classifier on MI355X.

Workload (BASELINE.json config 4 = config 1 replicated): `--loops` independent Branin BO
loops per GPU (default 512, the whole of config 4 on one GPU), 16-16-1 MLP, gamma 0.25,
fit(epochs=200, batch_size=64) warm-started every iteration, argmax with 3 L-BFGS-B
restarts from 1024 uniform samples (maxiter 1000, ftol 1e-9), 10 initial points.  One
"step" is one BO iteration of EVERY loop on the GPU; the data set grows by one point per
step.  Loops are sharded over ranks (weak scaling, no data-path collective; one gather of
the results at the end, outside the timed region).  `--schedule async` (default): every loop
advances on its own, one fused kernel (labels -> fit -> screen -> restarts -> pick) per
loop-iteration, launched in batches of whatever loops are ready; `--schedule groups`: four
groups of loops in lock-step, five launches per group-iteration.  Same trajectories.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement) with the
`roofline` of the dominant kernel (async: iteration_kernel, the only big one; groups: the one
with the largest share of the step, lbfgsb_kernel, `kernels` lists fit_kernel too;
algorithmic bytes per SURVEY.md §8d divided by the HIP-event duration measured on the
launching stream) and a `cpu_baseline` (the numpy/scipy oracle, which
mirrors the reference's per-step structure, timed on this host's cores -- one single-threaded
process per core -- for a bounded sample; `cpu_baseline_1core` is the one-core figure).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The replica engine steps groups of loops on separate HIP streams.  ROCm multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them taken by the null stream);
# streams sharing a queue serialise.  Must be set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0   # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md


def _branin01(X):
    """Branin-Hoo rescaled to [0, 1]^2 (same as bore_amd.engine.branin01; kept here so that the
    CPU workers do not import the GPU package)."""
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)


def _cpu_worker(job):
    """Oracle BO loops (config 1) on ONE host core for `seconds`: label -> per-step eager fit ->
    predict -> sequential scipy L-BFGS-B with one single-point f/g call per evaluation.
    Returns (BO iterations done, seconds)."""
    seconds, iters_per_loop, seed = job
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ[k] = "1"
    from scipy.optimize import Bounds
    from oracle import bore_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:          # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    with ctx:
        acts = ["relu", "relu", "sigmoid"]
        bounds = Bounds(np.zeros(2), np.ones(2))
        it, loops, t0 = 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:      # fresh loops of `iters_per_loop` iterations:
            rs = np.random.RandomState(seed + 1000 * loops)   # the same N range the GPU run covers
            p = O.glorot_uniform_params(2, [16, 16, 1], rs)
            st = O.AdamState(p)
            X = rs.uniform(size=(10, 2))
            y = _branin01(X)
            loops += 1
            for _ in range(iters_per_loop):
                z, _ = O.labels(y, 0.25)
                perms = np.stack([rs.permutation(len(y)) for _ in range(200)])
                O.fit(p, acts, st, X, z, perms, batch_size=64)
                res = O.argmax(p, acts, bounds, num_starts=3, num_samples=1024, random_state=rs)
                x = res.x if res is not None else rs.uniform(size=2)
                X = np.vstack([X, x])
                y = np.append(y, _branin01(x))
                it += 1
                if time.perf_counter() - t0 >= seconds:
                    break
        return it, time.perf_counter() - t0


def cpu_baseline(seconds, iters_per_loop=100, max_workers=64):
    """The CPU restatement of the reference's loop on the host cores of this box: independent
    BO loops, one single-threaded process per core (all cores, count stated), plus
    the one-core figure.  Bounded: `seconds` of wall time for the all-core run, a third of it
    for the one-core run."""
    import multiprocessing as mp
    cores = min(len(os.sched_getaffinity(0)), max_workers)
    it1, dt1 = _cpu_worker((seconds / 3.0, iters_per_loop, 0))
    what = ("numpy fp32 oracle + scipy L-BFGS-B (sequential restarts, single-point f/g), "
            f"loops of <= {iters_per_loop} iterations (N 10->{10 + iters_per_loop})")
    one = dict(value=it1 / dt1, unit="BO-iterations/s", cores=1, kind="port",
               sample=f"{it1} BO iterations in {dt1:.1f} s on one core; {what}")
    if cores <= 1:
        return one, None
    try:
        with mp.get_context("spawn").Pool(cores) as pool:   # (spawn: the parent holds a HIP context)
            t0 = time.perf_counter()
            res = pool.map_async(_cpu_worker, [(seconds, iters_per_loop, 17 + w)
                                               for w in range(cores)]).get(timeout=3 * seconds + 90)
            wall = time.perf_counter() - t0
    except Exception as e:                                   # pragma: no cover
        one["sample"] += f" (all-core run failed: {e})"
        return one, None
    its = sum(r[0] for r in res)
    rate = sum(r[0] / r[1] for r in res)          # steady state: process start-up not counted
    allc = dict(value=rate, unit="BO-iterations/s", cores=cores, kind="port",
                sample=f"{its} BO iterations by {cores} single-threaded processes (of "
                       f"{len(os.sched_getaffinity(0))} hardware threads available; one per core, "
                       f"independent loops) running {seconds:.0f} s each ({wall:.1f} s wall with "
                       f"start-up); {what}")
    return allc, one


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loops", type=int, default=512, help="BO loops per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--groups", type=int, default=4,
                    help="loop groups stepping on separate streams (overlaps L-BFGS-B tails)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank flow on a one-GPU box together with BORE_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--engine", default="native", choices=["native", "python"],
                    help="host loop of the replica engine: native = bore_engine_* (C++), python = "
                         "bore_amd.engine.ReplicaEngine (the same trajectories, bit for bit)")
    ap.add_argument("--schedule", default="async", choices=["groups", "async"],
                    help="native engine: groups = loop groups in lock-step on their own streams; "
                         "async = every loop re-enters the next launch as soon as its own restarts "
                         "are done (same trajectories)")
    ap.add_argument("--mode", default="device", choices=["device", "lockstep"],
                    help="device: L-BFGS-B restarts inside one kernel; lockstep: scipy on the host")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from bore_amd.engine import NativeEngine, ReplicaEngine, gather_results, shard_loop_ids

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if os.environ.get("BORE_BENCH_ONE_DEVICE") == "1":     # rehearsal: every rank on cuda:0
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    loop_ids = shard_loop_ids(rank, world, args.loops)     # contiguous shard per rank
    native = args.engine == "native" and args.mode == "device"
    if native:
        eng = NativeEngine(loop_ids, groups=args.groups, async_loops=args.schedule == "async")
    else:
        eng = ReplicaEngine(loop_ids, mode=args.mode, groups=args.groups)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    eng.run(args.warmup)
    if native:
        eng.take_stats(reset=True)
    else:
        eng.finish_timing()
        for k in ("fit_ms", "fit_bytes", "argmax_ms", "argmax_bytes"):
            eng.stats[k] = []
        eng.stats["n_fg_rows"] = eng.stats["n_rounds"] = 0
        eng.stats["host_enqueue_s"] = eng.stats["host_finalize_s"] = 0.0
        eng.stats["none_results"] = 0
    n_start = eng.N

    barrier()
    t0 = time.perf_counter()
    eng.run(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if native:
        st = eng.take_stats()
        n_groups = eng.n_groups
    else:
        eng.finish_timing()
        n_groups = len(eng.groups)
        st = dict(fit_ms=float(np.sum(eng.stats["fit_ms"])), fit_launches=len(eng.stats["fit_ms"]),
                  fit_bytes=float(np.sum(eng.stats["fit_bytes"], dtype=np.float64)),
                  argmax_ms=float(np.sum(eng.stats["argmax_ms"])),
                  argmax_launches=len(eng.stats["argmax_ms"]),
                  argmax_bytes=float(np.sum(eng.stats["argmax_bytes"], dtype=np.float64)),
                  n_fg_rows=eng.stats["n_fg_rows"], n_rounds=eng.stats["n_rounds"],
                  none_results=eng.stats["none_results"],
                  host_enqueue_s=eng.stats.get("host_enqueue_s", 0.0),
                  host_finalize_s=eng.stats.get("host_finalize_s", 0.0))

    tmax = torch.tensor([dt], dtype=torch.float64,
                        device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax[0])
    results = gather_results(eng, world)            # the path's only collective (RCCL gather)

    if rank == 0:
        total_iters = args.loops * world * args.steps

        # HBM traffic per launch from the committed PMC pass (FETCH_SIZE / WRITE_SIZE, gfx950
        # correction applied): it is per model, launches here carry loops/groups models
        try:
            with open(os.path.join(ROOT, "profiles", "r1", "pmc_traffic.json")) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
        models_per_launch = args.loops / n_groups if st["fit_ms"] else args.loops * args.steps / max(st["argmax_launches"], 1)

        def roof(name, ms_sum, bytes_sum, launches):
            # HIP-event durations (recorded on the launching stream) summed over the timed region
            ach = bytes_sum / (ms_sum * 1e-3) / 1e9
            per_model = pmc.get(name, {}).get("hbm_bytes_per_model")
            return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    "traffic": None if per_model is None else per_model * models_per_launch,
                    "avg_launch_ms": float(ms_sum / launches),
                    "algorithmic_bytes_per_launch": float(bytes_sum / launches),
                    "launches": int(launches),
                    # kernel-busy time / wall time = launches of this kernel in flight on average
                    # (stream groups / asynchronous batches overlap, so shares add up to > 1)
                    "share_of_step": float(ms_sum / (1e3 * dt)),
                    # all launches together: algorithmic bytes of the timed region / wall time
                    "aggregate_GBs": float(bytes_sum / dt / 1e9)}

        if st["fit_ms"] == 0.0:        # asynchronous schedule: fit + argmax are ONE kernel per launch
            kernels = [roof("iteration_kernel", st["argmax_ms"], st["fit_bytes"] + st["argmax_bytes"],
                            st["argmax_launches"])]
        else:
            kernels = [roof("fit_kernel", st["fit_ms"], st["fit_bytes"], st["fit_launches"])]
            if st["argmax_launches"]:
                kernels.append(roof("lbfgsb_kernel", st["argmax_ms"], st["argmax_bytes"],
                                    st["argmax_launches"]))
        dominant = max(kernels, key=lambda k: k["share_of_step"])
        # secondary figure: algorithmic FLOPs of the timed region against the fp32
        # vector/MFMA peak -- with theta in LDS the path is arithmetic/latency bound, not HBM bound
        D_, units_ = 2, [16, 16, 1]
        M_ = sum(a * b for a, b in zip([D_] + units_[:-1], units_))            # MACs per row
        P_ = M_ + sum(units_)
        Ns_ = np.arange(n_start, n_start + args.steps)
        rows = float(args.loops * world * 200 * Ns_.sum())                     # S_rows = E * N per fit
        adam = float(args.loops * world * 200 * np.ceil(Ns_ / 64).sum())       # S = E * ceil(N / B)
        flops = (rows * (6 * M_ - 2 * D_ * units_[0]) + adam * 12 * P_
                 + 2.0 * M_ * 1024 * total_iters + 4.0 * M_ * st["n_fg_rows"] * world)
        flop_roof = {"bound": "mfma", "what": "whole timed region, all kernels (secondary; SURVEY 8d)",
                     "achieved": flops / dt / 1e12, "peak": 157.3, "unit": "TFLOP/s",
                     "frac": flops / dt / 1e12 / 157.3,
                     "algorithmic_flops_per_iteration": flops / total_iters}
        out = {
            "metric": "BO-iterations/sec (fit+argmax), 16-16-1 MLP",
            "value": total_iters / dt, "unit": "BO-iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 4 (= config 1 x independent loops): Branin-2D, "
                                   "16-16-1 MLP, q=0.25, 200 epochs, batch 64, 3 L-BFGS-B "
                                   "restarts from 1024 samples",
                       "loops_per_gpu": args.loops, "restarts": args.mode,
                       "host_loop": "native" if native else "python",
                       "schedule": args.schedule if native else "groups",
                       "stream_groups": None if (native and args.schedule == "async") else n_groups,
                       "worker_streams": (int(os.environ.get("BORE_ASYNC_WORKERS", 12))
                                          if (native and args.schedule == "async") else None),
                       "N_start": int(n_start),
                       "N_end": int(eng.N), "parallelism": f"replica-shard x{world}"},
            "roofline": dominant,
            "roofline_flops": flop_roof,
            "kernels": kernels,
            "phases": {"fit_ms_per_launch": float(st["fit_ms"] / max(st["fit_launches"], 1)),
                       "fg_rows_per_step": st["n_fg_rows"] / args.steps,
                       "fg_rounds_per_step": st["n_rounds"] / args.steps,
                       "none_results": int(st["none_results"]),
                       "host_enqueue_ms_per_step": 1e3 * st["host_enqueue_s"] / args.steps,
                       "host_finalize_ms_per_step": 1e3 * st["host_finalize_s"] / args.steps},
            "best_y_median": float(np.median(results[:, -1])),
        }
        out["cpu_baseline"] = None
        if args.cpu_seconds > 0 and world == 1:              # (rank 0 at N = 1 only)
            allc, one = cpu_baseline(args.cpu_seconds, iters_per_loop=args.steps + args.warmup)
            out["cpu_baseline"] = allc
            if one is not None:
                out["cpu_baseline_1core"] = one
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
