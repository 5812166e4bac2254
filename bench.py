#!/usr/bin/env python
"""bench.py -- BO-iterations/sec (fit + argmax) for the 16-16-1 classifier on MI355X.

Workload (BASELINE.json config 4 = config 1 replicated): independent Branin BO loops, 16-16-1
MLP, gamma 0.25, fit(epochs=200, batch_size=64) warm-started every iteration, argmax with 3
L-BFGS-B restarts from 1024 uniform samples (maxiter 1000, ftol 1e-9), 10 initial points.  One
"step" is one BO iteration of EVERY loop of the job; the data set grows by one point per step.

Loops and GPUs.  Loops are sharded over ranks in contiguous blocks, no data-path collective; one
gather of the results at the end (RCCL), outside the timed region.
  --gpus 1                     512 loops on the GPU (all of config 4 on one GPU).
  --gpus N (N > 1), default    BASELINE config 4 AS WRITTEN: 512 loops in total, 512/N per GPU
                               ("scaling": "strong"; --total-loops T changes the total).
  --gpus N --loops L           L loops PER GPU ("scaling": "weak").
With N > 1 rank 0 afterwards times the single-GPU reference points alone (the other ranks wait)
and the line carries both efficiencies of SURVEY.md 8e:
  eff_w = T(N, total) / (N * T(1, total/N))   same per-GPU load   (what the >= 0.9 target uses)
  eff_s = T(N, total) / (N * T(1, total))     same total problem
`python bench.py --gpus N` from a plain shell starts its own N ranks (one process per GPU,
before anything in the parent touches HIP) and relays rank 0's line; under torchrun (RANK set)
it runs as the rank it is given.

Schedules: `--schedule async` (default): every loop advances on its own, one fused kernel
(labels -> fit -> screen -> restarts -> pick) per loop-iteration, launched in batches of whatever
loops are ready; `--schedule groups`: four groups of loops in lock-step, five launches per
group-iteration.  Same trajectories.

The LAST line of rank 0's stdout is the driver's record: one compact JSON line, never above 4 KB (`compact_line`),
with BASELINE.json's metric, the `roofline` of the dominant kernel (algorithmic bytes per SURVEY.md 8d divided by the
HIP-event duration measured on the launching stream), `roofline_flops` (secondary), `cpu_baseline` (the numpy/scipy
oracle, which mirrors the reference's per-step structure, on this host's cores -- one single-threaded process per core
-- for a bounded sample), the efficiencies for N > 1 and one figure per BASELINE config.  Everything else -- `runs`
(N = 1: the timed region is repeated on fresh engines while it is shorter than a second; `value` is the median run),
`configs` (per-phase times, algorithmic bytes / FLOPs, both roofline fractions, per-repetition evaluation counts and a
bounded CPU-oracle timing per BASELINE config), `kernels`, `phases`, `ranks` -- goes to the side file the record names
(`detail`: bench_detail_n{N}.json beside this script and under gpurun_out/).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# The replica engine runs independent launches on a dozen HIP streams.  ROCm multiplexes streams
# onto GPU_MAX_HW_QUEUES hardware queues (default 4, one of them taken by the null stream);
# streams sharing a queue serialise.  Must be set before the HIP runtime initialises (the engine
# checks what it got and says so in the line: config.hw_queues).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0    # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md
FP32_PEAK_TF = 157.3     # fp32 vector / MFMA peak, same guide
BF16_PEAK_TF = 2500.0    # dense bf16 MFMA peak
TOTAL_LOOPS = 512        # BASELINE.json config 4
def _latest(name):
    """profiles/rN/<name> of the newest round that has it."""
    for rnd in ("r6", "r5", "r4", "r3"):
        if os.path.exists(os.path.join(ROOT, "profiles", rnd, name)):
            return os.path.join("profiles", rnd, name)
    return os.path.join("profiles", "r6", name)


TRAFFIC_FILE = _latest("pmc_traffic.json")
# the optimiser's own float64 operations per evaluation request, counted in a host build of lbfgsb.h on device-trained
# networks (tools/lbfgsb_flops.py, tests/native/lbfgsb_flops.cpp)
OPT_FLOPS_FILE = _latest("optimiser_flops.json")
FP64_PEAK_TF = FP32_PEAK_TF / 2   # float64 vector rate: half the float32 vector rate of the same guide


def optimiser_fp64(name, requests, seconds):
    """Restart-phase roofline of the L-BFGS-B bookkeeping itself: counted float64 operations per evaluation request
    (a committed count, not taken in this run) x the run's requests over the phase's time, against the float64
    vector peak.  None when the count is missing."""
    try:
        with open(os.path.join(ROOT, OPT_FLOPS_FILE)) as f:
            e = json.load(f)["configs"][name]
    except Exception:
        return None
    per = e["addsubmul_per_evaluation"] + e["divsqrt_per_evaluation"]
    ach = per * requests / max(seconds, 1e-12) / 1e12
    return {"float64_ops_per_evaluation_request": per, "of_which_divisions_and_square_roots": e["divsqrt_per_evaluation"],
            "achieved_TFs": ach, "peak_TFs": FP64_PEAK_TF, "frac": ach / FP64_PEAK_TF,
            "network_flops_per_evaluation": e["network_flops_per_evaluation_fwd_bwd"],
            "source": f"{OPT_FLOPS_FILE}: lbfgsb.h's float64 operations counted in a host build, {e['restarts_counted']} "
                      "restarts of a device-trained loop; says the phase is a serial latency chain, not arithmetic"}
SWEEP_FILE = _latest("loops_sweep.json")   # T(1, L) measured on one GPU


def csrc_digest():
    """sha256 over the kernel sources: what a committed PMC pass was collected on (tools/collect_r6.py stores it in
    pmc_traffic.json) against what this run times -- `traffic_stale` in the line.  (bore_amd._lib.source_digest: the
    same digest the library carries.)"""
    from bore_amd import _lib
    return _lib.source_digest()


def predicted_efficiency(world, total):
    """What the committed single-GPU sweep T(1, L) says the N-GPU job will do when the ranks do not
    disturb each other: T(N, total) = N * T(1, total / N), hence eff_w = 1 and
    eff_s = T(1, total / N) / T(1, total).  None when the sweep lacks a point."""
    try:
        with open(os.path.join(ROOT, SWEEP_FILE)) as f:
            sw = json.load(f)
        t_share, t_all = sw["it_per_s"][str(total // world)], sw["it_per_s"][str(total)]
    except Exception:
        return None
    return {"eff_s": t_share / t_all, "eff_w": 1.0, "T_N_total": world * t_share,
            "gain_over_one_gpu": world * t_share / t_all,
            "source": f"{SWEEP_FILE} ({sw.get('build', '?')}): T(1,{total // world}) = {t_share:.0f}, "
                      f"T(1,{total}) = {t_all:.0f} BO-iterations/s; assumes ranks do not interact "
                      "(no data-path collective)"}


def _branin01(X):
    """Branin-Hoo rescaled to [0, 1]^2 (same as bore_amd.engine.branin01; kept here so that the
    CPU workers do not import the GPU package)."""
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)


# ------------------------------------------------------------------------------------------
# self-spawn: `python bench.py --gpus N` from a plain shell
# ------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n, argv):
    """Start n ranks of this script (children, NOT exec: this process never touches the GPU, the
    children initialise HIP themselves), wait for all, relay rank 0's stdout.  Returns the exit
    code: 0 only when every rank exited 0."""
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", LOCAL_WORLD_SIZE=str(n),
               MASTER_PORT=env.get("BORE_BENCH_PORT") or str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY=env.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv),
                                      env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # rank 0's stdout is drained by a thread while EVERY rank is polled: a rank that dies before a
    # rendezvous or barrier would otherwise leave rank 0 (and this parent) waiting for the
    # process-group timeout.  First non-zero exit -> the rest are killed; one overall deadline.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    deadline = time.time() + float(env.get("BORE_BENCH_DEADLINE_S", 1500))
    try:
        while True:
            codes = [p.poll() for p in procs]          # (every rank, every round)
            if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
                break
            if time.time() > deadline:
                rc = 124
                print("bench.py: ranks still running at the deadline: killed", file=sys.stderr)
                break
            time.sleep(0.05)
    finally:
        for p in procs:                      # a failed rank must not leave the others waiting
            if p.poll() is None:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f"bench.py: rank {r} exited with code {p.returncode}", file=sys.stderr)
            rc = rc or (p.returncode if p.returncode and p.returncode > 0 else 1)
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return rc


def pin_rank_cores(rank, world):
    """Give this rank its own slice of the host cores the container may use: the ranks of one node
    share a cgroup quota (16 cores for 8 ranks on the driver's box) and every rank runs a Python
    main thread plus the engine's service thread(s).  Returns the cores, or None when nothing was
    pinned (one rank, or an affinity mask that cannot be changed).  The slice is taken from the
    affinity mask in order, `usable / world` cores per rank where usable = min(mask, cgroup quota)."""
    if world <= 1:
        return None
    try:
        allowed = sorted(os.sched_getaffinity(0))
        n_mask, quota = _cpu_quota()
        usable = n_mask
        if quota:
            parts = quota.split()
            if parts[0] != "max" and int(parts[0]) > 0:
                period = int(parts[1]) if len(parts) > 1 else 100000
                usable = max(1, min(n_mask, int(parts[0]) // period))
        per = max(1, usable // world)
        start = (rank * per) % len(allowed)
        cores = [allowed[(start + i) % len(allowed)] for i in range(per)]
        os.sched_setaffinity(0, cores)
        return cores
    except (OSError, ValueError, AttributeError):
        return None


def rank_census(rank, world, local, backend, dry, cores):
    """What the backend itself says about the job: its world size, and every rank's device -- the
    proof in the N > 1 line that the collective backend (RCCL) saw N ranks on N different GPUs."""
    import torch
    import torch.distributed as dist
    me = {"rank": rank, "local_rank": local, "pid": os.getpid(), "host_cores": cores,
          "device": None if dry else torch.cuda.get_device_name(local),
          "device_index": None if dry else int(torch.cuda.current_device())}
    if not dry:
        try:
            pr = torch.cuda.get_device_properties(local)
            me["device_uuid"] = str(getattr(pr, "uuid", "")) or None
            me["pci_bus_id"] = getattr(pr, "pci_bus_id", None)
        except Exception:
            pass
    if world == 1:
        return {"backend": None, "backend_world_size": 1, "ranks": [me]}
    everyone = [None] * world
    dist.all_gather_object(everyone, me)
    return {"backend": dist.get_backend(), "backend_world_size": dist.get_world_size(), "ranks": everyone}


def resolve_loops(args, world):
    """(loops per GPU, total, scaling).  See the module docstring."""
    if args.loops is not None and args.total_loops is not None:
        raise SystemExit("bench.py: give --loops (per GPU) or --total-loops, not both")
    if args.loops is not None:
        return args.loops, args.loops * world, "weak"
    total = args.total_loops if args.total_loops is not None else TOTAL_LOOPS
    if total % world:
        raise SystemExit(f"bench.py: --total-loops {total} does not divide over {world} GPUs")
    return total // world, total, "weak" if world == 1 else "strong"


# ------------------------------------------------------------------------------------------
# CPU baseline (the oracle = the checker, timed beside the GPU path; never part of it)
# ------------------------------------------------------------------------------------------
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


def _cpu_quota():
    """What the container lets this process use: (hardware threads in the affinity mask, cgroup
    cpu.max as 'quota period' or None)."""
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                quota = f.read().strip()
            break
        except OSError:
            pass
    return len(os.sched_getaffinity(0)), quota


def cpu_baseline(seconds, iters_per_loop=100, max_workers=64):
    """The CPU restatement of the reference's loop on the host cores of this box: independent
    BO loops, one single-threaded process per core (all cores, count stated), plus
    the one-core figure.  Bounded: `seconds` of wall time for the all-core run, a third of it
    for the one-core run."""
    import multiprocessing as mp
    threads, quota = _cpu_quota()
    cores = min(threads, max_workers)
    try:        # cgroup v2 "quota period": more processes than the quota allows only throttle
        q, per = quota.split()
        if q != "max":
            cores = max(1, min(cores, int(-(-int(q) // int(per)))))
    except Exception:
        pass
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
                scaling_vs_1core=rate / (it1 / dt1),
                sample=f"{its} BO iterations by {cores} single-threaded processes (of {threads} "
                       f"hardware threads in the affinity mask; cgroup cpu.max = {quota!r} bounds the "
                       f"usable cores; one process per usable core, independent loops) running {seconds:.0f} s each ({wall:.1f} s "
                       f"wall with start-up); {what}")
    return allc, one


def tf_keras_probe():
    """the tf.keras CPU row exists only where TensorFlow imports."""
    import importlib.util
    try:
        found = importlib.util.find_spec("tensorflow") is not None
    except Exception:       # pragma: no cover
        found = False
    return "importable (row not implemented: never seen on this image)" if found else "unavailable"


# ------------------------------------------------------------------------------------------
# one timed region on one engine
# ------------------------------------------------------------------------------------------
class _DryEngine:
    """--dry-run: stands in for the replica engine so that the multi-rank flow of this script (rank
    spawning, rendezvous, barriers, max-over-ranks time, the gather) runs on a box without a GPU
    (tests/test_bench_spawn.py).  It computes nothing and its line says so."""

    def __init__(self, loop_ids):
        import torch
        self.loop_ids, self.N, self.n_groups = np.asarray(loop_ids, dtype=np.int64), 10, 1
        self.device = torch.device("cpu")

    def run(self, n):
        time.sleep(0.002 * n)
        self.N += n

    def take_stats(self, reset=True):
        z = dict.fromkeys(("fit_ms", "fit_bytes", "host_enqueue_s", "host_finalize_s"), 0.0)
        z.update(dict.fromkeys(("fit_launches", "n_fg_rows", "n_fg_requests", "n_rounds", "none_results"), 0))
        z.update(argmax_ms=1.0, argmax_bytes=1.0, argmax_launches=1)
        return z

    def best(self):
        L = len(self.loop_ids)
        return np.zeros((L, 2)), self.loop_ids.astype(np.float64)


def timed_run(args, loop_ids, barrier, steps=None, warmup=None):
    """W untimed warm-up steps, then EXACTLY K steps between barriers.  Returns a dict with the
    wall time, the engine's statistics of the timed region and the engine."""
    steps = args.steps if steps is None else steps
    warmup = args.warmup if warmup is None else warmup
    native = args.engine == "native" and args.mode == "device"
    shards = 1
    if args.dry_run:
        eng = _DryEngine(loop_ids)
    elif native:
        from bore_amd.engine import NativeEngine
        # (--objective native: the library's built-in Branin, evaluated by the engine's host loop itself;
        # python: the same function as a numpy callback through ctypes)
        kw = dict(groups=args.groups, async_loops=args.schedule == "async",
                  **({"objective": "branin01"} if args.objective == "native" else {}))
        # more loops than the device holds at once (> 512): the engine's work-queue schedule (one
        # persistent launch fed by the host).  --host-shards n > 1: the launch-per-batch schedule
        # instead, sharded over n engines / host threads (built-in objective only)
        shards = args.host_shards if args.host_shards > 0 else 1
        if shards > 1 and args.schedule == "async" and args.objective == "native":
            from bore_amd.engine import ShardedEngine
            eng = ShardedEngine(loop_ids, shards=shards, **kw)
        else:
            shards = 1
            eng = NativeEngine(loop_ids, **kw)
    else:
        from bore_amd.engine import ReplicaEngine
        eng = ReplicaEngine(loop_ids, mode=args.mode, groups=args.groups)
    eng.run(warmup)
    if native or args.dry_run:
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
    eng.run(steps)
    barrier()
    dt = time.perf_counter() - t0
    if native or args.dry_run:
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
    return dict(dt=dt, st=st, eng=eng, n_start=n_start, n_end=eng.N, n_groups=n_groups,
                native=native, loops=len(loop_ids), steps=steps, host_shards=shards)


# ------------------------------------------------------------------------------------------
# BASELINE configs 2 / 3 / 5 and single-loop config 1 (N = 1 only; `configs` in the line)
# ------------------------------------------------------------------------------------------
WIDE_CONFIGS = {
    # name: D, units, restarts R, screening samples Ns, data-set size N, arithmetic
    "cfg2_hartmann6_32-32-1_R256": dict(D=6, units=[32, 32, 1], R=256, Ns=1024, N=256,
                                        compute="float32"),
    "cfg3_hpo16_64-64-64-1_R1024": dict(D=16, units=[64, 64, 64, 1], R=1024, Ns=1024, N=256,
                                        compute="float32"),
    "cfg5_nas32_128-128-1_bf16_R4096": dict(D=32, units=[128, 128, 1], R=4096, Ns=4096, N=256,
                                            compute="bfloat16"),
}


# The network the reference's only in-repo caller builds (bore/plugins/hpbandster/base.py:23-33 -> DenseSequential's
# fall-through, bore/models.py:16-19): D -> 32-32-32-1, elu x3 + a linear output under from_logits BCE,
# transform="sigmoid", num_starts=5, num_samples=1024, gamma = 1/3, epochs = num_steps_per_iter // steps per epoch
# = 1000 // ceil(N / 64) (base.py:176-184).  Static shape 5: at 16 inputs as compiled, at 6 zero-padded (fit) /
# with the input dimension as a run-time argument (acquisition).
# (the third leg: the same network and fit with transform="identity" -- one of the plugin's own choices,
# plugins/hpbandster/base.py:18 -- whose acquisition surface is the logit itself: sigmoid(-f) of a 500-epoch classifier
# is flat at the screened starts, L-BFGS-B returns at iteration 0 and the default legs time no restart that iterates)
PLUGIN_CONFIGS = {
    "plugin_D16_transform_identity": dict(D=16, units=[32, 32, 32, 1], acts=["elu", "elu", "elu", "linear"],
                                          transform="identity", R=5, Ns=1024, N=100, gamma=1.0 / 3.0, epochs=1000 // 2,
                                          compute="float32"),
    "plugin_default_D6": dict(D=6, units=[32, 32, 32, 1], acts=["elu", "elu", "elu", "linear"], transform="sigmoid",
                              R=5, Ns=1024, N=100, gamma=1.0 / 3.0, epochs=1000 // 2, compute="float32"),
    "plugin_default_D16": dict(D=16, units=[32, 32, 32, 1], acts=["elu", "elu", "elu", "linear"], transform="sigmoid",
                               R=5, Ns=1024, N=100, gamma=1.0 / 3.0, epochs=1000 // 2, compute="float32"),
}


def _synthetic(rs, L, N, D):
    """Seeded smooth synthetic objective on [0,1]^D (value distribution irrelevant to cost)."""
    X = rs.uniform(size=(L, N, D))
    c = rs.uniform(0.2, 0.8, size=D)
    y = np.sum((X - c) ** 2, axis=2) + 0.1 * np.sin(5.0 * X.sum(axis=2))
    return X, y


def _counts(D, units):
    M = sum(a * b for a, b in zip([D] + units[:-1], units))
    return M, M + sum(units)


def config_gpu(name, c, loops, epochs=200, batch=64, reps=3):
    """`loops` independent models of config `c`: fit -> sample + screen -> R L-BFGS-B restarts ->
    pick, per phase HIP events on the launching stream.  One BO iteration of a loop = the
    sequence; loops are grid-parallel inside each launch."""
    import torch
    from bore_amd import _lib, ops
    D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
    acts = c.get("acts") or ["relu"] * (len(units) - 1) + ["sigmoid"]
    transform, gamma, epochs = c.get("transform", "identity"), c.get("gamma", 0.25), c.get("epochs", epochs)
    desc = _lib.make_desc(D, units, acts, compute=c["compute"])
    M, P = _counts(D, units)
    rs = np.random.RandomState(0)
    th = np.zeros((loops, P), dtype=np.float32)
    for l in range(loops):
        off, fan = 0, D
        for u in units:
            lim = np.sqrt(6.0 / (fan + u))
            th[l, off:off + fan * u] = rs.uniform(-lim, lim, size=fan * u)
            off += fan * u + u
            fan = u
    X, y = _synthetic(rs, loops, N, D)
    z = (y < np.quantile(y, gamma, axis=1)[:, None]).astype(np.float32)
    dev = torch.device("cuda", torch.cuda.current_device())
    theta = torch.from_numpy(th).to(dev)
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    t = torch.zeros(loops, dtype=torch.int64, device=dev)
    Xd, zd = torch.from_numpy(X.astype(np.float32)).to(dev), torch.from_numpy(z).to(dev)
    lo, hi = np.zeros(D), np.ones(D)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(reps + 1)]
    infos = []
    for k in range(reps + 1):                       # k = 0 warms up (first-launch set-up)
        e = ev[k]
        e[0].record()
        ops.mlp_fit(desc, theta, m, v, t, Xd, zd, epochs, batch, seed=0, epoch0=k * epochs,
                    want_loss=False)
        e[1].record()
        x0, _ = ops.sample_screen_topk(desc, theta, 0, Ns, lo, hi, R, draw_index=k)
        e[2].record()
        x, fun, jac, info = ops.lbfgsb_minimize(desc, theta, x0, lo, hi, transform, True,
                                                maxiter=1000, ftol=1e-9)
        e[3].record()
        ops.select_best(x, fun, info)
        e[4].record()
        infos.append(info)
    torch.cuda.synchronize()
    per_rep = np.array([[ev[k][i].elapsed_time(ev[k][i + 1]) for i in range(4)]
                        for k in range(1, reps + 1)])                     # [rep][phase] ms
    ph = np.median(per_rep, axis=0)                                       # ms per phase: the median rep
    nfev = np.stack([i.cpu().numpy()[:, :, 1] for i in infos[1:]]).astype(np.float64)
    nit = np.stack([i.cpu().numpy()[:, :, 0] for i in infos[1:]]).astype(np.float64)
    rows = float(nfev.sum() / reps)                                      # f/g rows per iteration
    rounds = float(nfev.max(axis=2).sum() / reps)                        # rounds, summed over loops
    status = np.stack([i.cpu().numpy()[:, :, 2] for i in infos[1:]])
    S = epochs * -(-N // batch)
    by = dict(fit=loops * epochs * (4.0 * N * (D + 1) + -(-N // batch) * 24.0 * P),
              screen=loops * (4.0 * Ns * (D + 1) + 4.0 * P),
              fg=rows * 4.0 * (2 * D + 1) + rounds * 4.0 * P)
    fl = dict(fit=loops * (epochs * N * (6.0 * M - 2.0 * D * units[0]) + S * 12.0 * P),
              screen=loops * 2.0 * M * Ns, fg=4.0 * M * rows)
    total_ms = float(ph.sum())
    peak_tf = BF16_PEAK_TF if c["compute"] == "bfloat16" else FP32_PEAK_TF
    kern = {"fit": ph[0], "screen": ph[1], "fg": ph[2]}
    # HBM traffic by the counters, per phase: the committed PMC pass of the 256-loop launches of this config
    # (tools/collect_r6.py: per kernel AND grid size).  Not collected in this run; stale when the sources moved.
    traffic = {}
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            pmc = json.load(f)
        if loops == 256:
            for k, e in pmc.get("configs", {}).get(name, {}).items():
                if isinstance(e, dict) and "hbm_bytes_per_launch" in e:
                    traffic[k] = dict(kernel=e.get("kernel"), bytes=e["hbm_bytes_per_launch"],
                                      frac_by_counters=e["hbm_bytes_per_launch"] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                      issue_slot_utilisation=e.get("issue_slot_utilisation"),
                                      stale=pmc.get("csrc_sha256") != csrc_digest())
    except Exception:
        pass
    return {
        "loops": loops, "it_per_s": loops / (total_ms * 1e-3), "epochs": int(epochs), "adam_steps": int(S),
        "ms": {"fit": float(ph[0]), "screen": float(ph[1]), "lbfgsb": float(ph[2]),
               "pick": float(ph[3]), "iteration": total_ms},
        # (the driver's box and the builder's differ by up to 20 % on these launches: every phase is
        # the median of `reps` repetitions -- each a new fit on top of the last, new starts -- with
        # the spread beside it)
        "ms_reps": {"n": int(reps), "value_is": "median",
                    "min": {k: float(per_rep[:, i].min()) for i, k in enumerate(("fit", "screen", "lbfgsb", "pick"))},
                    "max": {k: float(per_rep[:, i].max()) for i, k in enumerate(("fit", "screen", "lbfgsb", "pick"))}},
        # every repetition by itself: its restart launch's time beside the evaluation requests it served (a repetition
        # is a further fit and new starts: other surfaces, other counts) -- ms per million requests is the comparable figure
        "lbfgsb_reps": [{"ms": float(per_rep[k, 2]), "nfev": float(nfev[k].sum()), "nit": float(nit[k].sum()),
                         "nfev_max": float(nfev[k].max()),
                         "ms_per_million_requests": float(per_rep[k, 2] / max(nfev[k].sum(), 1.0) * 1e6)}
                        for k in range(reps)],
        "nit_per_restart": float(nit.mean()), "nfev_per_restart": float(nfev.mean()),
        "us_per_adam_step": 1e3 * float(ph[0]) / S,
        "fg_rows_per_iteration": rows, "restarts_ok_frac": float(np.mean(status <= 1)),
        "algorithmic_bytes": {k: float(x) for k, x in by.items()},
        "algorithmic_flops": {k: float(x) for k, x in fl.items()},
        # ("frac": the streaming MODEL's bytes over measured time -- may exceed 1, see the note; "traffic" /
        # "frac_by_counters": what the counters saw move, the fraction to quote)
        "roofline_hbm": {k: {"achieved_GBs": by[k] / (kern[k] * 1e-3) / 1e9,
                             "frac": by[k] / (kern[k] * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_is": "model bytes / time",
                             "traffic": traffic.get(k)} for k in by},
        "roofline_fma": {k: {"achieved_TFs": fl[k] / (kern[k] * 1e-3) / 1e12,
                             "frac": fl[k] / (kern[k] * 1e-3) / 1e12 / peak_tf,
                             "peak_TFs": peak_tf} for k in fl},
        # (the restart phase's OTHER arithmetic: the optimiser's float64 bookkeeping around the evaluations)
        "roofline_fp64_optimiser": optimiser_fp64(name, float(nfev.sum() / reps), kern["fg"] * 1e-3),
        # (SURVEY 8d's streaming model charges every Adam step a read and a write of theta, m and v: 24 P bytes.
        # The kernels keep theta in LDS or registers for the whole launch -- the bfloat16 fit its float32 masters
        # in registers, m and v in HBM: 16 P bytes per step -- so the fit's "achieved" is the model's bytes over
        # the measured time, not traffic, and can pass the peak.)
        "roofline_hbm_note": "algorithmic bytes of SURVEY 8d (streaming model) over measured time; theta stays on "
                             "chip for a launch, so the fit's fraction is not HBM traffic and may exceed 1",
    }


def config_cpu(c, epochs_sample=4, restarts_sample=6, batch=64, epochs=200):
    """The oracle on ONE host core for a bounded sample of config `c`: a few epochs of the fit, the
    screening forward and a few sequential scipy restarts, scaled to 200 epochs / R restarts."""
    from scipy.optimize import Bounds
    from oracle import bore_oracle as O
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:          # pragma: no cover
        import contextlib
        ctx = contextlib.nullcontext()
    D, units, R, Ns, N = c["D"], c["units"], c["R"], c["Ns"], c["N"]
    acts = c.get("acts") or ["relu"] * (len(units) - 1) + ["sigmoid"]
    transform, gamma, epochs = c.get("transform", "identity"), c.get("gamma", 0.25), c.get("epochs", epochs)
    restarts_sample = min(restarts_sample, R)
    rs = np.random.RandomState(0)
    with ctx:
        p = O.glorot_uniform_params(D, units, rs)
        st = O.AdamState(p)
        X, y = _synthetic(rs, 1, N, D)
        z, _ = O.labels(y[0], gamma)
        perms = np.stack([rs.permutation(N) for _ in range(epochs_sample)])
        bf16 = c["compute"] == "bfloat16"
        t0 = time.perf_counter()
        (O.fit_bf16 if bf16 else O.fit)(p, acts, st, X[0], z, perms, batch_size=batch)
        t_fit = (time.perf_counter() - t0) * epochs / epochs_sample
        t0 = time.perf_counter()
        res = O.maxima(p, acts, Bounds(np.zeros(D), np.ones(D)), num_starts=restarts_sample,
                       num_samples=Ns, random_state=rs, transform=transform)
        t_arg = time.perf_counter() - t0
        t0 = time.perf_counter()
        O.predict(p, acts, rs.uniform(size=(Ns, D)))
        t_scr = time.perf_counter() - t0
    t_restart = max(t_arg - t_scr, 0.0) / restarts_sample
    total = t_fit + t_scr + t_restart * R
    return dict(value=1.0 / total, unit="BO-iterations/s", cores=1, kind="port",
                ms=dict(fit=1e3 * t_fit, screen=1e3 * t_scr, lbfgsb=1e3 * t_restart * R),
                # (context only: the restarts run on a net that has seen `epochs_sample` epochs, not `epochs` --
                # another objective surface, other evaluation counts than the device's, whose restarts follow
                # the full fit; the fit and screening legs are like for like)
                same_work_as_device=False,
                sample=f"{epochs_sample} of {epochs} epochs of the fit and {restarts_sample} of {R} "
                       f"sequential scipy restarts (fp32 numpy arithmetic"
                       f"{'; the fit in the bf16 statement' if bf16 else ''}), scaled; mean nit "
                       f"{np.mean([r.nit for r in res]):.1f}, nfev {np.mean([r.nfev for r in res]):.1f}")


def _synthetic_objective(X):
    """The smooth synthetic objective of _synthetic (centre 0.4), as the engine's callback: [n, D] -> [n]."""
    return np.sum((X - 0.4) ** 2, axis=-1) + 0.1 * np.sin(5.0 * X.sum(axis=-1))


def plugin_engine(loops, steps=8, warmup=2):
    """The network the reference's only in-repo caller builds (PLUGIN_CONFIGS: 16 -> 32-32-32-1, elu, transform sigmoid,
    5 restarts from 1024 samples, gamma 1/3, 500 epochs at N ~ 100) as whole BO LOOPS on the replica engine: round 6
    gives static shape 5 the fused loop kernel (label -> fit -> screen -> restarts -> pick, resident workgroups, two
    loops per CU; beyond that the work queue), where round 5 ran four lock-step launches per batch.  One step = one BO
    iteration of every loop; the objective is a numpy callback."""
    import torch
    from bore_amd.engine import NativeEngine
    c = PLUGIN_CONFIGS["plugin_default_D16"]
    eng = NativeEngine(np.arange(loops), input_dim=c["D"], units=tuple(c["units"]), acts=tuple(c["acts"]),
                       transform=c["transform"], gamma=c["gamma"], epochs=c["epochs"], num_starts=c["R"],
                       num_samples=c["Ns"], n_init=c["N"], objective=_synthetic_objective, async_loops=True)
    eng.run(warmup)
    eng.take_stats(reset=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.take_stats()
    n_it = max(st.get("phase_iterations", 0), 1)
    out = {"loops": loops, "steps": steps, "it_per_s": loops * steps / dt, "ms": {"iteration": 1e3 * dt / steps},
           "N_start": int(c["N"] + warmup), "N_end": int(eng.N), "schedule": "fused loop kernel" if st["fit_ms"] == 0.0 else "launch chain",
           "loops_per_cu": st.get("loops_per_cu"), "side_by_side_workgroups": st.get("side_by_side_workgroups"),
           "per_loop_iteration_us": {k: 1e-3 * st["phase_ns_" + k] / n_it for k in ("labels", "fit", "screen", "lbfgsb")},
           "none_results": int(st["none_results"]), "fg_requests_per_iteration": st.get("n_fg_requests", 0) / n_it}
    eng.close()
    return out


def all_configs(args, barrier, cpu):
    out = {}
    # config 1 as ONE loop (the literal "Branin-2D ... 3 restarts" configuration)
    r = timed_run(args, np.arange(1, dtype=np.int64), barrier, steps=max(args.steps, 20), warmup=3)
    out["cfg1_branin2_16-16-1_single_loop"] = {
        "loops": 1, "it_per_s": r["steps"] / r["dt"], "ms": {"iteration": 1e3 * r["dt"] / r["steps"]},
        "N_start": int(r["n_start"]), "N_end": int(r["n_end"]),
        "note": "chain latency of one loop: own fit + own restarts + host hand-over"}
    if hasattr(r["eng"], "close"):
        r["eng"].close()
    del r
    for name, c in list(WIDE_CONFIGS.items()) + list(PLUGIN_CONFIGS.items()):
        try:
            one = config_gpu(name, c, loops=1)
            many = config_gpu(name, c, loops=256, reps=5)
            out[name] = dict(one, many_loops=many)
            if cpu:
                out[name]["cpu_baseline"] = config_cpu(c)
        except Exception as e:                     # a config that cannot run says so in the line
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
    for loops in (256, 512):                       # (whole loops of the plugin's network on the replica engine)
        try:
            out[f"plugin_default_D16_engine_{loops}_loops"] = plugin_engine(loops)
        except Exception as e:
            out[f"plugin_default_D16_engine_{loops}_loops"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


# ------------------------------------------------------------------------------------------
def run_rank(args):
    import torch
    import torch.distributed as dist
    from bore_amd.engine import gather_results, shard_loop_ids

    dry = args.dry_run
    if dry:
        args.backend = "gloo"
    sync = (lambda: None) if dry else torch.cuda.synchronize
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("BORE_BENCH_FAIL_RANK") == str(rank):    # (tests: a rank that dies early)
        raise SystemExit(f"bench.py: rank {rank} told to fail (BORE_BENCH_FAIL_RANK)")
    if os.environ.get("BORE_BENCH_ONE_DEVICE") == "1":     # rehearsal: every rank on cuda:0
        local = 0
    # (the affinity mask and the cgroup quota are the NODE's: slice them by this node's ranks -- torchrun and
    # spawn_ranks export LOCAL_RANK / LOCAL_WORLD_SIZE -- not by the job's)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    cores = pin_rank_cores(int(os.environ.get("LOCAL_RANK", rank)), local_world)     # (before the engine starts its threads: they inherit the mask)
    if not dry:
        torch.cuda.set_device(local)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group("gloo")
    census = rank_census(rank, world, local, args.backend, dry, cores)
    loops, total, scaling = resolve_loops(args, world)

    def barrier():
        sync()
        if world > 1:
            dist.barrier()
        sync()

    def solo_barrier():
        sync()

    loop_ids = shard_loop_ids(rank, world, loops)          # contiguous shard per rank
    t_flow0 = time.perf_counter()
    r = timed_run(args, loop_ids, barrier)
    tmax = torch.tensor([r["dt"], time.perf_counter() - t_flow0], dtype=torch.float64,
                        device="cuda" if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    wall_first = float(tmax[1])                     # engine creation + warm-up + timed steps
    runs = [dict(value=total * args.steps / float(tmax[0]), dt=float(tmax[0]), r=r)]
    results = gather_results(r["eng"], world)       # the path's only collective (RCCL gather)
    if hasattr(r["eng"], "close"):
        r["eng"].close()
    r["eng"] = None

    # A timed region of K steps is tens of milliseconds here: one sample of a noisy quantity.  The
    # region (W warm-up + EXACTLY K timed steps, barriers on both sides, max over ranks) is repeated
    # on fresh engines until --min-timed-s of timed region have accumulated; `value` is the median
    # run, the line carries quartiles / min / max.  The repeat count follows from the first run's
    # max-over-ranks time, so every rank does the same number.
    if args.repeats != 1:
        if args.repeats > 0:
            n_rep = args.repeats
        else:
            n_rep = int(min(args.max_repeats, max(1, np.ceil(args.min_timed_s / max(runs[0]["dt"], 1e-6)))))
            # (bounded by wall time as well: the driver gives the whole command a few minutes)
            n_rep = int(max(1, min(n_rep, args.repeat_budget_s // max(wall_first, 1e-3))))
        for _ in range(n_rep - 1):
            rr = timed_run(args, loop_ids, barrier)
            if hasattr(rr["eng"], "close"):
                rr["eng"].close()
            rr["eng"] = None
            tm = torch.tensor([rr["dt"]], dtype=torch.float64,
                              device="cuda" if args.backend == "nccl" and not dry else "cpu")
            if world > 1:
                dist.all_reduce(tm, op=dist.ReduceOp.MAX)
            runs.append(dict(value=total * args.steps / float(tm[0]), dt=float(tm[0]), r=rr))
    order = sorted(range(len(runs)), key=lambda i: runs[i]["value"])
    med = runs[order[(len(runs) - 1) // 2]]         # (lower median for an even count)
    r, dt = med["r"], med["dt"]
    st = r["st"]

    t_repeats_done = time.perf_counter()
    # N > 1, config 4 as written (512 loops over the node): the same line also carries the WEAK point -- 512 loops on
    # EVERY GPU, what a single GPU runs for the headline -- so that one invocation per N gives the scaling curve that
    # means something for this path (eff_w: same per-GPU load) beside the as-written one (64 loops per GPU at N = 8:
    # a loop is a serial chain that an emptier GPU runs no faster).  All ranks take part; max over ranks.
    weak = None
    if world > 1 and not args.no_efficiency and args.loops is None and args.total_loops is None:
        rw = timed_run(args, shard_loop_ids(rank, world, TOTAL_LOOPS), barrier)
        if hasattr(rw["eng"], "close"):
            rw["eng"].close()
        tw = torch.tensor([rw["dt"]], dtype=torch.float64, device="cuda" if args.backend == "nccl" and not dry else "cpu")
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        weak = {"loops_per_gpu": TOTAL_LOOPS, "total_loops": TOTAL_LOOPS * world, "scaling": "weak",
                "value": TOTAL_LOOPS * world * args.steps / float(tw[0]), "ms_per_step": 1e3 * float(tw[0]) / args.steps}
        del rw
    # N > 1: the single-GPU reference points of the two efficiencies, timed by rank 0 ALONE
    eff = None
    if world > 1 and not args.no_efficiency:
        if rank == 0:
            # (medians of repeated regions, like the N-GPU figure itself)
            n_ref = int(max(1, min(len(runs), 21)))

            def ref_point(n_loops):
                vs = []
                for _ in range(n_ref):
                    q = timed_run(args, shard_loop_ids(0, 1, n_loops), solo_barrier)
                    vs.append(n_loops * args.steps / q["dt"])
                    if hasattr(q["eng"], "close"):
                        q["eng"].close()
                    del q
                return float(np.median(vs))

            v_share = ref_point(loops)
            v_all = ref_point(total)
            v = med["value"]
            if weak is not None:          # (T(1, 512) is v_all when the job is config 4 as written)
                weak["T_1"] = v_all if total == TOTAL_LOOPS else ref_point(TOTAL_LOOPS)
                weak["eff_w"] = weak["value"] / (world * weak["T_1"])
            eff = {"T_N_total": v, "T_1_share": v_share, "T_1_total": v_all,
                   "share_loops": loops, "total_loops": total,
                   "eff_w": v / (world * v_share), "eff_s": v / (world * v_all),
                   "definitions": "eff_w = T(N,total)/(N*T(1,total/N)); eff_s = T(N,total)/(N*T(1,total)) "
                                  "(SURVEY.md 8e); reference points timed by rank 0 alone in this "
                                  f"invocation, same steps/warm-up, median of {n_ref} fresh engines each",
                   "target_0.9_uses": "eff_w (same per-GPU load; the path has no collective, so what "
                                      "it measures is host contention between the ranks). eff_s is "
                                      "bounded by one loop's dependency chain: a GPU with total/N "
                                      "loops runs each of them no faster than with all of them"}
        barrier()
    t_eff_done = time.perf_counter()

    if rank == 0:
        try:
            with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
                pmc = json.load(f)
        except Exception:
            pmc = {}
        digest = csrc_digest()
        models_per_launch = (loops / r["n_groups"] if st["fit_ms"]
                             else loops * args.steps / max(st["argmax_launches"], 1))

        def roof(name, ms_sum, bytes_sum, launches):
            # HIP-event durations (recorded on the launching stream) summed over the timed region
            ach = bytes_sum / (ms_sum * 1e-3) / 1e9
            per_model = pmc.get(name, {}).get("hbm_bytes_per_model")
            return {"bound": "hbm", "kernel": name, "achieved": ach, "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                    # (not measured in this run: the committed PMC pass, per model x models per launch)
                    "traffic": None if per_model is None else per_model * models_per_launch,
                    "traffic_source": None if per_model is None else
                    f"{TRAFFIC_FILE} ({pmc.get('build', '?')}; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                    "passes, committed; NOT collected in this run)",
                    # the committed pass belongs to THESE kernel sources (hash stored at collection)?
                    "traffic_stale": None if per_model is None else pmc.get("csrc_sha256") != digest,
                    # the bound that means something for a chain of dependent small steps: wave-cycles in which a
                    # wave issued an instruction / all wave-cycles (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES of the same
                    # committed pass; its complement is mostly SQ_WAIT_ANY)
                    "issue_slot_utilisation": pmc.get(name, {}).get("issue_slot_utilisation"),
                    "waiting_share_of_wave_cycles": pmc.get(name, {}).get("waiting_share_of_wave_cycles"),
                    "avg_launch_ms": float(ms_sum / launches),
                    "algorithmic_bytes_per_launch": float(bytes_sum / launches),
                    "launches": int(launches),
                    # kernel-busy time / wall time = launches of this kernel in flight on average
                    # (stream groups / asynchronous batches overlap, so shares add up to > 1)
                    "share_of_step": float(ms_sum / (1e3 * dt)),
                    # all launches together: algorithmic bytes of the timed region / wall time
                    "aggregate_GBs": float(bytes_sum / dt / 1e9),
                    "aggregate_frac": float(bytes_sum / dt / 1e9 / HBM_PEAK_GBS)}

        if st["fit_ms"] == 0.0:        # asynchronous schedule: fit + argmax are ONE kernel per launch
            kernels = [roof("iteration_kernel", st["argmax_ms"], st["fit_bytes"] + st["argmax_bytes"],
                            st["argmax_launches"])]
            # (only while every loop of the launch is RESIDENT -- as many workgroups side by side as loops, the
            # engine says: with more loops than the device holds at once -- the work-queue schedule, which is ONE
            # launch per run as well -- workgroups serve several loops each, busy time per launch is no longer
            # the sum / loops and the figures would overstate the rate)
            side = int(st.get("side_by_side_workgroups") or 0)
            if st.get("phase_iterations") and st["argmax_launches"] <= 4 and side >= loops:
                # A resident launch spans the host's turn-around as well: its workgroups wait on their
                # CUs for the objective values.  `achieved` / `frac` above are per the contract (a
                # launch's algorithmic bytes / its HIP-event duration, = the rocprofv3 duration); the
                # figures below take the waiting out: the in-kernel clock stamps give every
                # loop-iteration's busy time (labels + fit + screen + restarts), and a launch's
                # workgroups run side by side, so busy time per launch = that sum / loops in flight.
                k0 = kernels[0]
                busy_s = 1e-9 * sum(st["phase_ns_" + q] for q in ("labels", "fit", "screen", "lbfgsb"))
                per_launch_busy_s = busy_s / max(loops, 1) / max(st["argmax_launches"], 1)
                ach_busy = (st["fit_bytes"] + st["argmax_bytes"]) / max(st["argmax_launches"], 1) \
                    / max(per_launch_busy_s, 1e-12) / 1e9
                k0["device_busy_ms_per_launch"] = 1e3 * per_launch_busy_s
                k0["waiting_share_of_launch"] = max(0.0, 1.0 - 1e3 * per_launch_busy_s / k0["avg_launch_ms"])
                k0["achieved_busy"] = ach_busy
                k0["frac_busy"] = ach_busy / HBM_PEAK_GBS
                k0["note"] = ("avg_launch_ms is the HIP-event span of a resident launch: device work plus "
                              "on-CU waiting for the host's objective values and the tail of the slowest "
                              "loop; *_busy use the in-kernel phase stamps instead (mean loop)")
        else:
            kernels = [roof("fit_kernel", st["fit_ms"], st["fit_bytes"], st["fit_launches"])]
            if st["argmax_launches"]:
                kernels.append(roof("lbfgsb_kernel", st["argmax_ms"], st["argmax_bytes"],
                                    st["argmax_launches"]))
        dominant = max(kernels, key=lambda k: k["share_of_step"])
        # secondary figure: algorithmic FLOPs of the timed region against the fp32
        # vector/MFMA peak -- with theta in LDS the path is arithmetic/latency bound, not HBM bound
        D_, units_ = 2, [16, 16, 1]
        M_, P_ = _counts(D_, units_)
        Ns_ = np.arange(r["n_start"], r["n_start"] + args.steps)
        rows = float(loops * 200 * Ns_.sum())                        # S_rows = E * N per fit
        adam = float(loops * 200 * np.ceil(Ns_ / 64).sum())          # S = E * ceil(N / B)
        flops = (rows * (6 * M_ - 2 * D_ * units_[0]) + adam * 12 * P_
                 + 2.0 * M_ * 1024 * loops * args.steps + 4.0 * M_ * st["n_fg_rows"])   # rank 0's
        flop_roof = {"bound": "mfma", "what": "whole timed region of one GPU, all kernels (secondary; SURVEY 8d)",
                     "achieved": flops / dt / 1e12, "peak": FP32_PEAK_TF, "unit": "TFLOP/s",
                     "frac": flops / dt / 1e12 / FP32_PEAK_TF,
                     "algorithmic_flops_per_iteration": flops / (loops * args.steps)}
        # (beside the network's FLOPs: the float64 bookkeeping of the restarts' optimiser, counted -- what the restart
        # phase, half of a loop-iteration, actually computes with)
        flop_roof["optimiser_fp64"] = optimiser_fp64("cfg1_branin2_16-16-1_R3",
                                                     float(st.get("n_fg_requests", st["n_fg_rows"])), dt)
        phases = {"fg_rows_per_step": st["n_fg_rows"] / args.steps,          # evaluations that ran the network
                  # (the optimisers' nfev: also counts trial points the image shortcut served)
                  "fg_requests_per_step": st.get("n_fg_requests", st["n_fg_rows"]) / args.steps,
                  "fg_rounds_per_step": st["n_rounds"] / args.steps,
                  "none_results": int(st["none_results"]),
                  "host_enqueue_ms_per_step": 1e3 * st["host_enqueue_s"] / args.steps,
                  "host_finalize_ms_per_step": 1e3 * st["host_finalize_s"] / args.steps}
        if st["fit_ms"]:
            phases["fit_ms_per_launch"] = float(st["fit_ms"] / max(st["fit_launches"], 1))
        if st.get("phase_iterations"):     # fused kernel: in-kernel clock stamps, mean per loop-iteration
            n_it = st["phase_iterations"]
            phases["per_loop_iteration_us"] = {k: 1e-3 * st["phase_ns_" + k] / n_it
                                               for k in ("labels", "fit", "screen", "lbfgsb")}
            # the host's view of the same loop-iterations (means): a loop is ready (its new row
            # is known) -> its launch is enqueued -> the host sees its result -> objective done
            phases["per_loop_iteration_us"].update(
                host_ready_to_launch=1e6 * st["ready_to_launch_s"] / n_it,
                host_launch_to_result=1e6 * st["launch_to_result_s"] / n_it,
                host_result_to_ready=1e6 * st["result_to_ready_s"] / n_it)
            phases["loops_per_launch"] = n_it / max(st["batches"], 1)
        out = {
            "metric": "BO-iterations/sec (fit+argmax), 16-16-1 MLP",
            "value": med["value"], "unit": "BO-iterations/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 4 (= config 1 x independent loops): Branin-2D, "
                                   "16-16-1 MLP, q=0.25, 200 epochs, batch 64, 3 L-BFGS-B "
                                   "restarts from 1024 samples",
                       "loops_per_gpu": loops, "total_loops": total, "restarts": args.mode,
                       "host_loop": "native" if r["native"] else "python",
                       # the synthetic objective (evaluated on the host, excluded from the metric)
                       "objective": ("Branin-Hoo on [0, 1]^2, the library's built-in (C, inside the host loop)"
                                     if r["native"] and args.objective == "native" else
                                     "Branin-Hoo on [0, 1]^2, a numpy callback"),
                       "schedule": args.schedule if r["native"] else "groups",
                       "stream_groups": None if (r["native"] and args.schedule == "async") else r["n_groups"],
                       # engines (host threads) this GPU's loops are split over (> 1 only beyond 512 loops)
                       "host_shards": r.get("host_shards", 1),
                       "worker_streams": st.get("worker_streams"),
                       # (measured by the engine at creation: streams the device ran at once)
                       "stream_concurrency": st.get("stream_concurrency"),
                       "GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES"),
                       # how long a loop's workgroup waits on its CU for the objective value
                       # before it gives the slot up (0 = one launch per loop-iteration)
                       "resident_wait_us": 2000.0,      # (bore_engine_cfg::resident_wait_us: the library's default)
                       "N_start": int(r["n_start"]),
                       "N_end": int(r["n_end"]), "parallelism": f"replica-shard x{world}"},
            "roofline": dominant,
            "roofline_flops": flop_roof,
            "kernels": kernels,
            "phases": phases,
            "runs": {"values": [round(q["value"], 1) for q in runs], "n": len(runs),
                     "min": min(q["value"] for q in runs), "max": max(q["value"] for q in runs),
                     "p25": float(np.percentile([q["value"] for q in runs], 25)),
                     "p75": float(np.percentile([q["value"] for q in runs], 75)),
                     "iqr": float(np.subtract(*np.percentile([q["value"] for q in runs], [75, 25]))),
                     "value_is": "the median run" if len(runs) > 1 else "the only run",
                     "timed_region_s": dt,
                     "timed_total_s": float(sum(q["dt"] for q in runs)),
                     "each_run": f"fresh engine, {args.warmup} warm-up + exactly {args.steps} timed steps "
                                 "between barriers"},
            "best_y_median": float(np.median(results[:, -1])),
            "tf_keras": tf_keras_probe(),
            # the collective backend's own count of ranks and every rank's device and host cores
            "ranks": census,
        }
        out["flow_wall_s"] = {"repeated_timed_regions": t_repeats_done - t_flow0,
                              "efficiency_reference_runs": t_eff_done - t_repeats_done,
                              "bounds": f"repeats <= min({args.max_repeats}, {args.repeat_budget_s:.0f} s / "
                                        f"first run's {wall_first:.2f} s); 2 reference runs of the same "
                                        "steps on rank 0; then (N = 1 only) survey form, configs, CPU sample"}
        # T(1, L) of the committed single-GPU sweep (tools/profile_r4.sh): what one GPU does with more loops than
        # config 4 names -- the regime in which several GPUs pay (beyond 512 loops: the work-queue schedule)
        try:
            with open(os.path.join(ROOT, SWEEP_FILE)) as f:
                sw = json.load(f)
            out["loops_sweep_committed"] = {"it_per_s": sw["it_per_s"], "source": SWEEP_FILE,
                                            "stale": sw.get("csrc_sha256") != digest}
        except Exception:
            out["loops_sweep_committed"] = None
        if eff is not None:
            out["efficiency"] = eff
        if weak is not None:
            out["weak_point_512_loops_per_gpu"] = weak
        if world > 1:
            # said before it is measured: config 4 AS WRITTEN (512 loops in total) leaves each GPU
            # total/N loops, and a loop is a sequential chain that runs no faster on an emptier GPU
            out["efficiency_predicted"] = predicted_efficiency(world, total)
        out["cpu_baseline"] = None
        if dry:
            out["dry_run"] = ("no GPU work was done: this line only shows that the multi-rank flow "
                              "of bench.py ran; every figure in it is meaningless")
            out["value"] = None
        if world == 1 and not dry:
            if args.survey_steps > 0:
                # SURVEY.md 8d "Config 1" as written (T = 100 iterations after 3 warm-up, N 13 -> 113,
                # two batches per epoch from N = 65 on), x 512 loops: the same engine, longer loops
                sv = []
                for _ in range(3):
                    q = timed_run(args, loop_ids, solo_barrier, steps=args.survey_steps, warmup=3)
                    sv.append((total * args.survey_steps / q["dt"], q["dt"], q["n_start"], q["n_end"]))
                    if hasattr(q["eng"], "close"):
                        q["eng"].close()
                    q["eng"] = None
                sv.sort()
                out["survey_form"] = {"steps": args.survey_steps, "warmup": 3, "value": sv[1][0],
                                      "unit": "BO-iterations/s", "values": [round(x[0], 1) for x in sv],
                                      "ms_per_step": 1e3 * sv[1][1] / args.survey_steps,
                                      "N_start": int(sv[1][2]), "N_end": int(sv[1][3]),
                                      "value_is": "median of 3 fresh engines"}
            if not args.no_configs:
                out["configs"] = all_configs(args, solo_barrier, cpu=args.cpu_seconds > 0)
            if args.cpu_seconds > 0:                             # (rank 0 at N = 1 only)
                allc, one = cpu_baseline(args.cpu_seconds, iters_per_loop=args.steps + args.warmup)
                out["cpu_baseline"] = allc
                if one is not None:
                    out["cpu_baseline_1core"] = one
        emit(out, args)
    if world > 1:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------
# the line: compact record last on stdout (<= LINE_LIMIT bytes), everything else in a side file
# ------------------------------------------------------------------------------------------
LINE_LIMIT = 4096        # the driver keeps an 8 KB tail of stdout: the record must fit it with room to spare


def _sig(x, n=6):
    """Floats to n significant digits (the line is read by people and a parser, not re-computed from)."""
    if isinstance(x, float):
        return float(f"{x:.{n}g}") if np.isfinite(x) else None
    if isinstance(x, dict):
        return {k: _sig(v, n) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, n) for v in x]
    if isinstance(x, np.generic):
        return _sig(x.item(), n)
    return x


def _pick(d, keys):
    return None if d is None else {k: d[k] for k in keys if k in d and d[k] is not None}


def compact_line(out, detail_file):
    """The driver's record: BASELINE.json's metric with `roofline` and `cpu_baseline`, the efficiencies for N > 1
    and one figure per BASELINE config; `detail` names the file with everything else.  Never above LINE_LIMIT bytes:
    optional blocks are dropped, least important first, until it fits."""
    c = out.get("config") or {}
    line = {k: out.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = _pick(c, ("workload", "loops_per_gpu", "total_loops", "schedule", "host_loop", "N_start",
                               "N_end", "parallelism"))
    line["roofline"] = _pick(out.get("roofline"), (
        "bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "issue_slot_utilisation",
        "avg_launch_ms", "algorithmic_bytes_per_launch", "launches", "frac_busy"))
    line["roofline_flops"] = _pick(out.get("roofline_flops"), ("bound", "achieved", "peak", "unit", "frac"))
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if cb is None else dict(
        _pick(cb, ("value", "unit", "cores", "kind")), sample=str(cb.get("sample", ""))[:160])
    if out.get("cpu_baseline_1core"):
        line["cpu_baseline_1core"] = _pick(out["cpu_baseline_1core"], ("value", "cores", "kind"))
    if out.get("dry_run"):
        line["dry_run"] = True
    optional = []            # (key, value) in order of importance; dropped from the END when the line is too long
    if out.get("efficiency"):
        optional.append(("efficiency", _pick(out["efficiency"], (
            "eff_w", "eff_s", "T_N_total", "T_1_share", "T_1_total", "share_loops", "total_loops"))))
    if out.get("weak_point_512_loops_per_gpu"):
        optional.append(("weak_point_512_loops_per_gpu", _pick(out["weak_point_512_loops_per_gpu"], (
            "value", "ms_per_step", "loops_per_gpu", "total_loops", "scaling", "T_1", "eff_w"))))
    if out.get("efficiency_predicted"):
        optional.append(("efficiency_predicted", _pick(out["efficiency_predicted"], ("eff_w", "eff_s", "gain_over_one_gpu"))))
    if out.get("runs"):
        optional.append(("runs", _pick(out["runs"], ("n", "min", "p25", "p75", "max", "value_is"))))
    if out.get("ranks"):
        optional.append(("ranks", _pick(out["ranks"], ("backend", "backend_world_size"))))
    if out.get("configs"):
        cf = {}
        for name, e in out["configs"].items():
            if "error" in e:
                cf[name] = {"error": e["error"][:80]}
                continue
            m = e.get("many_loops", e)
            cf[name] = {"loops": m.get("loops"), "it_per_s": m.get("it_per_s"),
                        "ms": _pick(m.get("ms"), ("fit", "screen", "lbfgsb", "iteration"))}
        optional.append(("configs", cf))
    if out.get("survey_form"):
        optional.append(("survey_form", _pick(out["survey_form"], ("value", "steps", "N_start", "N_end"))))
    if out.get("phases"):
        ph = out["phases"]
        optional.append(("phases", {"none_results": ph.get("none_results"),
                                    "per_loop_iteration_us": _pick(ph.get("per_loop_iteration_us"),
                                                                   ("labels", "fit", "screen", "lbfgsb"))}))
    line["tf_keras"] = out.get("tf_keras")
    line["detail"] = detail_file
    for k, v in optional:
        line[k] = v
    line = _sig(line)
    while optional and len(json.dumps(line, separators=(",", ":"))) > LINE_LIMIT - 64:
        line.pop(optional.pop()[0])
    text = json.dumps(line, separators=(",", ":"))
    if len(text) > LINE_LIMIT:          # (cannot happen with the fixed key sets above; refuse rather than print it)
        raise SystemExit(f"bench.py: compact record is {len(text)} bytes (> {LINE_LIMIT})")
    return text


def emit(out, args):
    """Side file first (everything: `configs`, `kernels`, `phases`, `runs`, `ranks`, ...), then the compact record as
    the LAST line of stdout.  The side file goes beside the script (bench_detail_n{N}.json) and, when that directory
    exists, under gpurun_out/ as well so that a gpurun call brings it back."""
    name = f"bench_detail_n{out.get('n_gpus', 1)}.json"
    written = None
    for d in ([args.detail_dir] if args.detail_dir else [ROOT, os.path.join(ROOT, "gpurun_out")]):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, name), "w") as f:
                    json.dump(out, f)
                written = written or os.path.relpath(os.path.join(d, name), ROOT)
        except OSError:
            pass
    if args.print_detail:
        print(json.dumps(out))
    sys.stdout.flush()
    print(compact_line(out, written))
    sys.stdout.flush()


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--loops", type=int, default=None,
                    help="BO loops PER GPU (weak scaling); default: see --total-loops")
    ap.add_argument("--total-loops", type=int, default=None,
                    help=f"BO loops of the whole job, sharded over the GPUs (default {TOTAL_LOOPS} = "
                         "BASELINE config 4)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0,
                    help="wall seconds of the all-core CPU-oracle sample (a third of it on one core "
                         "first); 0 = no cpu_baseline")
    ap.add_argument("--repeats", type=int, default=0,
                    help="timed regions on fresh engines (0 = as many as --min-timed-s asks for)")
    ap.add_argument("--min-timed-s", type=float, default=1.5,
                    help="repeat the timed region until this much of it has accumulated")
    ap.add_argument("--max-repeats", type=int, default=80)
    ap.add_argument("--repeat-budget-s", type=float, default=120.0,
                    help="wall-time bound of the repeated timed regions (engine creation included)")
    ap.add_argument("--survey-steps", type=int, default=100,
                    help="N = 1: also time SURVEY.md 8d's form of config 1 x 512 (T = 100 after 3 "
                         "warm-up steps, N 13 -> 113) -> `survey_form` (0 = skip)")
    ap.add_argument("--no-configs", action="store_true",
                    help="skip the per-config figures (configs 1-single-loop, 2, 3, 5)")
    ap.add_argument("--dry-run", action="store_true",
                    help="no GPU: run the multi-rank flow (spawn, rendezvous over gloo, barriers, "
                         "gather) around an engine stand-in; the line carries \"dry_run\" and no value")
    ap.add_argument("--detail-dir", default=None,
                    help="directory of the side file bench_detail_n{N}.json (default: beside this script and gpurun_out/)")
    ap.add_argument("--print-detail", action="store_true",
                    help="also print the full record as a line BEFORE the compact one (the compact record stays last)")
    ap.add_argument("--no-efficiency", action="store_true",
                    help="N > 1: skip rank 0's single-GPU reference runs (eff_w / eff_s)")
    ap.add_argument("--groups", type=int, default=4,
                    help="loop groups stepping on separate streams (overlaps L-BFGS-B tails)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1 (nccl = RCCL; gloo only to rehearse the "
                         "multi-rank flow on a one-GPU box together with BORE_BENCH_ONE_DEVICE=1)")
    ap.add_argument("--engine", default="native", choices=["native", "python"],
                    help="host loop of the replica engine: native = bore_engine_* (C++), python = "
                         "bore_amd.engine.ReplicaEngine (the same trajectories, bit for bit)")
    ap.add_argument("--host-shards", type=int, default=0,
                    help="n > 1: the launch-per-batch schedule sharded over n engines / host threads (more than 512 "
                         "loops, asynchronous schedule, built-in objective); default: one engine (work queue beyond 512 loops)")
    ap.add_argument("--objective", default="native", choices=["native", "python"],
                    help="the synthetic Branin objective: the library's built-in (evaluated inside the "
                         "engine's host loop) or the same function as a numpy callback")
    ap.add_argument("--schedule", default="async", choices=["groups", "async"],
                    help="native engine: groups = loop groups in lock-step on their own streams; "
                         "async = every loop re-enters the next launch as soon as its own restarts "
                         "are done (same trajectories)")
    ap.add_argument("--mode", default="device", choices=["device", "lockstep"],
                    help="device: L-BFGS-B restarts inside one kernel; lockstep: scipy on the host")
    return ap.parse_args(argv)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse_args(argv)
    if args.gpus > 1 and "RANK" not in os.environ:
        # a plain shell: start the ranks ourselves.  Nothing above imported torch or touched HIP.
        resolve_loops(args, args.gpus)          # (argument errors before any process starts)
        return spawn_ranks(args.gpus, argv)
    run_rank(args)
    return 0


if __name__ == "__main__":
    sys.exit(main())
