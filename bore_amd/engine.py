"""Many independent BO loops on one GPU (BASELINE.json config 4).

One BO loop is a strictly sequential chain (iteration i+1 needs i; Adam step k+1 needs
k), so a single loop can never fill an MI355X.  Hyper-parameter-search replicas are
independent, though: loop l has its own seed, data set, classifier and Adam state.
``ReplicaEngine`` advances L such loops in lock-step, one BO iteration at a time, each
stage being ONE launch over the leading replica dimension:

    label   z = y < quantile(y, gamma)                (bore/data.py:31-35)
    fit     L models x epochs x ceil(N/B) Adam steps   (README.rst:93)       1 launch
    screen  predict on num_samples uniform candidates (bore/mixins.py:49-56) 1 launch
    argmax  L x num_starts L-BFGS-B restarts          (bore/mixins.py:57-89) 1 launch / round

Across GPUs the loops are sharded in contiguous blocks (rank r owns loops
[r*L, (r+1)*L): the device streams are keyed by first id + offset); no
data-path collective exists -- results are gathered once at the end (``gather_results``).
"""
from __future__ import annotations

import time

import ctypes as _lib_ctypes

import numpy as np
import torch

from . import _lib, ops
from .optimizers import lockstep
from .transforms import resolve


def branin01(X):
    """Branin-Hoo on its native box x1 in [-5, 10], x2 in [0, 15], rescaled to [0, 1]^2."""
    x1, x2 = 15.0 * X[..., 0] - 5.0, 15.0 * X[..., 1]
    return ((x2 - 5.1 / (4 * np.pi ** 2) * x1 ** 2 + 5 / np.pi * x1 - 6) ** 2
            + 10 * (1 - 1 / (8 * np.pi)) * np.cos(x1) + 10)


def initial_state(loop_ids, D, units, P, n_init, objective, low, high):
    """What every loop starts from: its numpy RandomState(loop id) -- the stream the reference
    would consume --, Keras-default weights (glorot_uniform kernels, zero biases) and n_init
    uniform observations, drawn from that stream in this order."""
    rss = [np.random.RandomState(int(s)) for s in loop_ids]
    th = np.empty((len(rss), P), dtype=np.float32)
    for i, rs in enumerate(rss):
        off, fan_in = 0, D
        for u in units:
            lim = np.sqrt(6.0 / (fan_in + u))
            th[i, off:off + fan_in * u] = rs.uniform(-lim, lim, size=fan_in * u)
            off += fan_in * u
            th[i, off:off + u] = 0.0
            off += u
            fan_in = u
    X0 = np.stack([rs.uniform(low, high, size=(n_init, D)) for rs in rss])
    return rss, th, X0, objective(X0)


LBFGSB_DEFAULTS = dict(maxcor=10, ftol=2.2204460492503131e-09, gtol=1e-5, maxfun=15000,
                       maxiter=15000, maxls=20)


def lbfgsb_opts(options):
    o = dict(LBFGSB_DEFAULTS)
    unknown = set(options) - set(o)
    if unknown:
        raise TypeError(f"unknown L-BFGS-B options: {sorted(unknown)}")
    o.update(options)
    return _lib.LbfgsbOpts(int(o["maxcor"]), int(o["maxiter"]), int(o["maxfun"]), int(o["maxls"]),
                           float(o["ftol"]), float(o["gtol"]))


# engines whose close() came while some engine of this process was inside run() (see NativeEngine.close)
_RUN_DEPTH = [0]
_DEFERRED = []
import threading as _threading_run
_RUN_LOCK = _threading_run.Lock()       # (ShardedEngine runs its engines on several threads)


class NativeEngine:
    """The replica engine with its host loop in C++ (``bore_engine_*``, bore_amd/csrc/
    bore_engine.hip): same loops, same streams of random numbers, same kernels and the same
    trajectories as ``ReplicaEngine(mode="device", select="device")`` -- bit for bit, tested --
    without the interpreter between a group's results and its next launch.  ``objective`` is called
    with [n, D] points (one group's suggestions) and returns n values."""

    def __init__(self, loop_ids, input_dim=2, units=(16, 16, 1), acts=("relu", "relu", "sigmoid"),
                 transform="identity", gamma=0.25, epochs=200, batch_size=64, num_starts=3,
                 num_samples=1024, n_init=10, objective=branin01, options=None, device=None,
                 seed=0, groups=4, deduplicate=False, async_loops=False, resident_wait_us=None,
                 worker_streams=None, work_queue=None):
        """async_loops: every loop advances on its own (a loop re-enters the next launch as soon
        as ITS restarts are done instead of waiting for the slowest loop of its group); same
        trajectories, tested.  Needs num_starts <= 16.  Its knobs (None = the library's default; bore_engine_cfg,
        ABI 12 -- environment variables until round 5): ``resident_wait_us`` how long a loop's workgroup waits on its
        CU for the objective value before it parks (0: one launch per loop-iteration), ``worker_streams``,
        ``work_queue`` False / True: never / always the persistent work-queue launch (default: by size)."""
        self.device = device or _lib.require_gpu()
        self.loop_ids = np.asarray(loop_ids, dtype=np.int64)
        self.L = L = len(self.loop_ids)
        assert L >= 1 and np.array_equal(self.loop_ids, self.loop_ids[0] + np.arange(L)), \
            "loop ids must be consecutive (the device streams are keyed by first id + offset)"
        self.D = D = int(input_dim)
        self.units, self.acts = list(units), list(acts)
        self.desc = _lib.make_desc(D, self.units, self.acts)
        self.P = ops.param_count(self.desc)
        tr = resolve(transform).negated()
        assert tr.name is not None, "the native engine runs on the device only: a named transform ('identity', 'sigmoid', 'exp')"
        assert tr.negate, "the engine minimises transform(-f(x)) (bore/mixins.py:20)"
        # objective="branin01": the library's built-in Branin (bore_objective_branin01: the same fp64
        # expression, evaluated by the host loop itself -- no interpreter between a result and the next
        # row; what bench.py times).  A callable is called back through ctypes as before.
        native = None
        if isinstance(objective, str):
            if objective != "branin01":
                raise ValueError(f"built-in objectives: 'branin01' (got {objective!r}); or pass a callable")
            assert D == 2, "branin01 is two-dimensional"
            native, objective = "bore_objective_branin01", branin01
        self.objective = objective
        self.low, self.high = np.zeros(D), np.ones(D)
        rss, th, X0, y0 = initial_state(self.loop_ids, D, self.units, self.P, n_init, objective,
                                        self.low, self.high)
        mt = np.empty((L, 625), dtype=np.uint32)
        for i, rs in enumerate(rss):
            _, key, pos, _, _ = rs.get_state()
            mt[i, :624], mt[i, 624] = key, pos
        self._error = None

        def _objective(xp, n, d, yp, _user):
            try:
                X = np.ctypeslib.as_array(xp, shape=(n, d))
                np.ctypeslib.as_array(yp, shape=(n,))[:] = self.objective(X)
                return 0
            except BaseException as e:          # never unwind through the C frames
                self._error = e
                return 1

        self._cb = _lib.OBJECTIVE_FN(_objective)
        if native is not None:
            self._cb = _lib_ctypes.cast(getattr(_lib.lib(), native), _lib.OBJECTIVE_FN)
        self._lo, lo_p = ops._host_f64(self.low, D, "low")
        self._hi, hi_p = ops._host_f64(self.high, D, "high")
        cfg = _lib.EngineCfg(L, max(1, min(int(groups), L)), int(self.loop_ids[0]), int(n_init),
                             int(epochs), int(batch_size), int(num_starts), int(num_samples),
                             _lib.TRANSFORM[tr.name], int(bool(deduplicate)), int(bool(async_loops)),
                             int(seed) & (2 ** 64 - 1), float(gamma),
                             _lib.AdamCfg(1e-3, 0.9, 0.999, 1e-7),
                             lbfgsb_opts(dict(options or dict(maxiter=1000, ftol=1e-9))), lo_p, hi_p,
                             -1 if resident_wait_us is None else int(resident_wait_us),
                             0 if worker_streams is None else int(worker_streams),
                             -1 if work_queue is None else int(bool(work_queue)), 0)
        self.n_groups = cfg.groups
        th = np.ascontiguousarray(th)
        X0, y0 = np.ascontiguousarray(X0, dtype=np.float64), np.ascontiguousarray(y0, dtype=np.float64)
        self._h = _lib_ctypes.c_void_p()
        vp = lambda a: a.ctypes.data_as(_lib_ctypes.c_void_p)
        torch.cuda.set_device(self.device)
        _lib.check(_lib.lib().bore_engine_create(_lib_ctypes.byref(self.desc), _lib_ctypes.byref(cfg),
                                                 vp(th), vp(X0), vp(y0), vp(mt), self._cb, None,
                                                 _lib_ctypes.byref(self._h)))

    def close(self):
        """Free the engine's device memory, pinned memory and streams now.  (The callback closure refers
        back to the engine: without this an engine lives until the cycle collector runs, and the streams of
        several dead engines share the process's hardware queues with the live one's.)

        NOT while any engine of the process is inside ``run``: ``bore_engine_destroy`` frees device memory, which waits
        for the device to be idle -- and a running engine's resident / work-queue kernel waits on its CUs for objective
        values that the host, stuck in the free, would never deliver (found in round 6: the cycle collector finalised a
        dead engine from INSIDE another engine's objective callback and the run hung).  Such a close is deferred to the
        end of the outermost ``run``."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            with _RUN_LOCK:
                defer = _RUN_DEPTH[0] > 0
                if defer:
                    _DEFERRED.append(h)
            if not defer:
                _lib.lib().bore_engine_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:       # (interpreter shutdown: the module's globals are gone; the process ends anyway)
            pass

    def run(self, n_steps):
        with _RUN_LOCK:
            _RUN_DEPTH[0] += 1
        try:
            rc = _lib.lib().bore_engine_run(self._h, int(n_steps))
        finally:
            with _RUN_LOCK:
                _RUN_DEPTH[0] -= 1
                late = [_DEFERRED.pop() for _ in range(len(_DEFERRED))] if _RUN_DEPTH[0] == 0 else []
            for h in late:                      # (engines closed or collected while a run was in progress)
                _lib.lib().bore_engine_destroy(h)
        if self._error is not None:             # the objective raised: the engine stopped
            e, self._error = self._error, None
            raise e
        _lib.check(rc)

    @property
    def N(self):
        return int(_lib.lib().bore_engine_size(self._h))

    def observations(self):
        N = self.N
        X, y = np.empty((self.L, N, self.D)), np.empty((self.L, N))
        vp = lambda a: a.ctypes.data_as(_lib_ctypes.c_void_p)
        _lib.check(_lib.lib().bore_engine_observations(self._h, vp(X), vp(y)))
        return X, y

    @property
    def X(self):
        return self.observations()[0]

    @property
    def y(self):
        return self.observations()[1]

    def state(self):
        """(theta, adam_m, adam_v [L, P] float32, adam_t [L] int64) as host arrays."""
        th, m, v = (np.empty((self.L, self.P), dtype=np.float32) for _ in range(3))
        t = np.empty(self.L, dtype=np.int64)
        vp = lambda a: a.ctypes.data_as(_lib_ctypes.c_void_p)
        _lib.check(_lib.lib().bore_engine_state(self._h, vp(th), vp(m), vp(v), vp(t)))
        return th, m, v, t

    def take_stats(self, reset=True):
        st = _lib.EngineStats()
        _lib.check(_lib.lib().bore_engine_get_stats(self._h, _lib_ctypes.byref(st), int(reset)))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def best(self):
        X, y = self.observations()
        i = np.argmin(y, axis=1)
        return X[np.arange(self.L), i], y[np.arange(self.L), i]




class ShardedEngine:
    """(Round 4's first answer to more loops than the device holds at once; the engine's work-queue
    schedule -- one persistent launch fed by the host, bore_iter.hip: queue_kernel -- has since become the
    default there and is faster.  Kept: it shards the launch-per-batch schedule.)
    One host thread serving all loops is the bound from ~2 048 loops on (result polling, bookkeeping,
    launches: ~1.3 us per loop-iteration).  ``shards`` NativeEngines over contiguous ranges of the loop ids, each driven
    by its own host thread (``bore_engine_run`` releases the interpreter lock; engines are independent
    of each other, include/bore_hip.h).  Meant for the built-in objective (``objective="branin01"``): a
    Python callback serialises the threads on the interpreter lock again.  A loop's trajectory does
    not depend on its shard (the streams are keyed by global loop id).

    The engines share one GPU: none of them may keep its workgroups resident (each would count the
    whole device as its own), and the worker streams are divided among them."""

    def __init__(self, loop_ids, shards=2, **kw):
        ids = np.asarray(loop_ids, dtype=np.int64)
        shards = int(max(1, min(shards, len(ids))))
        if shards > 1:
            # (no resident workgroups: each engine would count the whole device as its own; no work queue: its
            # kernel would hold the device until the run ends; the worker streams divided among the shards)
            kw = dict(kw, resident_wait_us=0, worker_streams=max(2, 12 // shards), work_queue=False)
        self.engines = [NativeEngine(part, **kw) for part in np.array_split(ids, shards)]
        e0 = self.engines[0]
        self.loop_ids, self.L, self.D, self.P = ids, len(ids), e0.D, e0.P
        self.n_groups = e0.n_groups
        self.device = e0.device

    def run(self, n_steps):
        import threading
        errors = []

        def work(eng):
            try:
                torch.cuda.set_device(eng.device)
                eng.run(n_steps)
            except BaseException as err:        # re-raised in the caller's thread
                errors.append(err)

        threads = [threading.Thread(target=work, args=(e,)) for e in self.engines[1:]]
        for t in threads:
            t.start()
        work(self.engines[0])
        for t in threads:
            t.join()
        if errors:
            raise errors[0]

    def close(self):
        for e in self.engines:
            e.close()

    @property
    def N(self):
        return self.engines[0].N

    def observations(self):
        parts = [e.observations() for e in self.engines]
        return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])

    @property
    def X(self):
        return self.observations()[0]

    @property
    def y(self):
        return self.observations()[1]

    def state(self):
        parts = [e.state() for e in self.engines]
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(4))

    def take_stats(self, reset=True):
        """Sums over the shards (worker_streams too; stream_concurrency: the smallest seen)."""
        stats = [e.take_stats(reset) for e in self.engines]
        out = {k: sum(s[k] for s in stats) for k in stats[0]}
        out["stream_concurrency"] = min(s["stream_concurrency"] for s in stats)
        out["loops_per_cu"] = min(s["loops_per_cu"] for s in stats)
        return out

    def best(self):
        X, y = self.observations()
        i = np.argmin(y, axis=1)
        return X[np.arange(self.L), i], y[np.arange(self.L), i]


class ReplicaEngine:
    """The Python statement of the replica engine (``NativeEngine`` runs the same loop in C++ and
    is what ``bench.py`` uses): kept as the readable reference and as the check of the native one."""

    def __init__(self, loop_ids, input_dim=2, units=(16, 16, 1), acts=("relu", "relu", "sigmoid"),
                 transform="identity", gamma=0.25, epochs=200, batch_size=64, num_starts=3,
                 num_samples=1024, n_init=10, objective=branin01, max_points=None,
                 options=None, device=None, mode="device", seed=0, groups=1, select="device",
                 deduplicate=False):
        """mode "device": label, fit, candidate draw, screening and all L-BFGS-B restarts run
        as five launches per BO iteration with ONE host sync (candidates from the device
        counter stream).  mode "lockstep": candidates from each loop's numpy RandomState and
        SciPy's L-BFGS-B on the host around the batched f/g kernel (the reference's streams
        and optimiser code; slow -- one launch + L*R host state machines per round).

        groups > 1 (device mode): the loops are split into that many contiguous groups, each
        stepping on its own HIP stream.  A BO iteration of a group ends when its SLOWEST
        L-BFGS-B restart does; with several groups in flight one group's tail overlaps the
        others' fit/argmax instead of idling the GPU (loops never interact, so grouping only
        changes scheduling -- every loop's trajectory is the same; tested).

        select "device" (device mode): the observations live on the device in fp64
        (``ops.ObservationStore``: one [L, D+1] upload per iteration instead of the whole data
        set) and ``bore_select_best`` picks each loop's suggestion there; "host": the numpy
        statement of the same rule on downloaded results (kept as the check of the device path).
        deduplicate: reject results that ``Record.is_duplicate`` finds among the loop's
        observations, as the plugin's filter_fn does (bore/plugins/hpbandster/base.py:210-214;
        the README loop has no filter -- the default)."""
        assert mode in ("device", "lockstep")
        assert select in ("device", "host")
        self.select, self.deduplicate = select, bool(deduplicate)
        self.mode, self.seed = mode, int(seed)
        self.device = device or _lib.require_gpu()
        self.loop_ids = np.asarray(loop_ids, dtype=np.int64)
        self.L = L = len(self.loop_ids)
        assert L >= 1 and np.array_equal(self.loop_ids, self.loop_ids[0] + np.arange(L)), \
            "loop ids must be consecutive (the device streams are keyed by first id + offset)"
        self.D = D = int(input_dim)
        self.units, self.acts = list(units), list(acts)
        self.desc = _lib.make_desc(D, self.units, self.acts)
        self.P = ops.param_count(self.desc)
        self.transform = resolve(transform)
        assert self.transform.name is not None, "the replica engine runs on the device only: a named transform"
        self.gamma, self.epochs, self.batch_size = gamma, int(epochs), int(batch_size)
        self.num_starts, self.num_samples = int(num_starts), int(num_samples)
        self.objective = objective
        self.options = dict(options or dict(maxiter=1000, ftol=1e-9))
        self.low, self.high = np.zeros(D), np.ones(D)
        self.rs, th, X0, y0 = initial_state(self.loop_ids, D, self.units, self.P, n_init,
                                            self.objective, self.low, self.high)
        self.theta = torch.from_numpy(th).to(self.device)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_t = torch.zeros(L, dtype=torch.int64, device=self.device)
        self._lo, self._lo_p = ops._host_f64(self.low, D, "low")
        self._hi, self._hi_p = ops._host_f64(self.high, D, "high")
        self._adam = _lib.AdamCfg(1e-3, 0.9, 0.999, 1e-7)
        self._lopts = lbfgsb_opts(self.options)
        G = max(1, min(int(groups), L)) if mode == "device" else 1
        bounds = np.linspace(0, L, G + 1).astype(int)
        self.groups = [_Group(int(a), int(b), X0[a:b], y0[a:b], self)
                       for a, b in zip(bounds[:-1], bounds[1:])]
        self.stats = dict(fit_ms=[], fit_bytes=[], argmax_ms=[], argmax_bytes=[], n_fg_rows=0,
                          n_rounds=0, none_results=0)
        self._ev, self._ev2 = [], []

    @property
    def N(self):
        return self.groups[0].X.shape[1]

    @property
    def X(self):
        """[L, N, D] observations (all groups are at the same iteration between run() calls)."""
        return np.concatenate([g.X for g in self.groups], axis=0)

    @property
    def y(self):
        return np.concatenate([g.y for g in self.groups], axis=0)

    # -- stages ------------------------------------------------------------
    def label(self):
        y = self.groups[0].y
        tau = np.quantile(y, q=self.gamma, axis=1)
        return np.less(y, tau[:, None])

    def fit(self, z):
        L, N, D = self.L, self.N, self.D
        g0 = self.groups[0]
        Xd = torch.from_numpy(g0.X.astype(np.float32)).to(self.device)
        zd = torch.from_numpy(z.astype(np.float32)).to(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mlp_fit(self.desc, self.theta, self.adam_m, self.adam_v, self.adam_t, Xd, zd,
                    self.epochs, self.batch_size, seed=self.seed,
                    model_index0=int(self.loop_ids[0]), epoch0=g0.epochs_seen, want_loss=False)
        e1.record()
        self._ev.append((e0, e1))
        steps = -(-N // self.batch_size)
        self.stats["fit_bytes"].append(L * self.epochs * (4 * N * (D + 1) + steps * 24 * self.P))
        g0.epochs_seen += self.epochs

    def screen(self):
        Xs = np.stack([rs.uniform(self.low, self.high, size=(self.num_samples, self.D))
                       for rs in self.rs])
        pred = ops.mlp_forward(self.desc, self.theta,
                               torch.from_numpy(Xs.astype(np.float32)).to(self.device))
        f_init = -pred.cpu().numpy()
        ind = np.argpartition(f_init, kth=self.num_starts - 1, axis=1)[:, :self.num_starts]
        return np.take_along_axis(Xs, ind[:, :, None], axis=1)       # [L, R, D]

    def restarts(self, X0):
        """L x R L-BFGS-B problems in lock-step; one f/g launch per round over all of them."""
        L, R, D = X0.shape
        buf = np.array(X0, dtype=np.float64).reshape(L * R, D)
        tr = self.transform.negated()

        def fg(Xp, idx):
            buf[idx] = Xp
            val, grad = ops.mlp_value_and_input_grad(
                self.desc, self.theta, torch.from_numpy(buf.reshape(L, R, D)).to(self.device),
                tr.name, tr.negate)
            self.stats["n_fg_rows"] += len(idx)
            self.stats["n_rounds"] += 1
            return val.cpu().numpy().reshape(-1)[idx], grad.cpu().numpy().reshape(L * R, D)[idx]

        res = lockstep.minimize_lockstep(fg, buf.copy(), bounds=list(zip(self.low, self.high)),
                                         with_index=True, **self.options)
        return [res[l * R:(l + 1) * R] for l in range(L)]

    def suggest(self, results):
        x_next = np.empty((self.L, self.D))
        for l, rl in enumerate(results):
            best = None
            for r in rl:
                if (r.success or r.status == 1) and (best is None or r.fun < best.fun):
                    best = r
            if best is None:     # reference: fall back to a random point
                self.stats["none_results"] += 1
                x_next[l] = self.rs[l].uniform(self.low, self.high)
            else:
                x_next[l] = best.x
        return x_next

    def _enqueue(self, g):
        """Queue one BO iteration of group g on its stream: 5-7 launches + async copies.  Every
        buffer is preallocated (an allocation or a pageable copy here would serialise the
        streams) and the C-ABI is called directly (shapes were validated when the group's
        buffers were made)."""
        t_host0 = time.perf_counter()
        Lg, N, D, R = g.b - g.a, g.X.shape[1], self.D, self.num_starts
        lib, C, ptr = _lib.lib(), _lib_ctypes, _lib.ptr
        dev_sel = self.select == "device"
        if dev_sel:
            if g.store.n + 1 > g.store.cap:
                g.store.grow(2 * g.store.cap)
                g.z_dev = torch.empty(Lg * g.store.cap, dtype=torch.float32, device=self.device)
            if g.store.n == N - 1:          # the newest row is still on the host
                g.new_x_pin_np[:] = g.X[:, -1]
                g.new_y_pin_np[:] = g.y[:, -1]
        else:
            if N > g.cap:
                g.alloc_inputs(self, max(2 * g.cap, N))
            # stage the observations in pinned memory, contiguous [Lg, N, D] / [Lg, N]
            np.copyto(g.X_pin_np[:Lg * N * D].reshape(Lg, N, D), g.X, casting="same_kind")
            np.copyto(g.y_pin_np[:Lg * N].reshape(Lg, N), g.y)
        th, m, v, t = (x[g.a:g.b] for x in (self.theta, self.adam_m, self.adam_v, self.adam_t))
        tr = self.transform.negated()
        with torch.cuda.stream(g.stream):
            sp = _lib.stream_ptr()
            if dev_sel:
                st = g.store
                if st.n == N - 1:
                    g.new_x.copy_(g.new_x_pin, non_blocking=True)
                    g.new_y.copy_(g.new_y_pin, non_blocking=True)
                    _lib.check(lib.bore_append_observations(
                        Lg, D, ptr(st.X), ptr(st.y), st.n, st.cap, ptr(g.new_x), ptr(g.new_y),
                        ptr(st.X32), ptr(st.y_dense), sp))
                    st.n += 1
                assert st.n == N
                g.X_dev, g.y_dev = st.X32, st.y_dense
            else:
                g.X_dev[:Lg * N * D].copy_(g.X_pin[:Lg * N * D], non_blocking=True)
                g.y_dev[:Lg * N].copy_(g.y_pin[:Lg * N], non_blocking=True)
            _lib.check(lib.bore_labels(Lg, ptr(g.y_dev), N, float(self.gamma), ptr(g.z_dev),
                                       None, sp))
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(lib.bore_mlp_fit(C.byref(self.desc), Lg, ptr(th), ptr(m), ptr(v), ptr(t),
                                        ptr(g.X_dev), ptr(g.z_dev), N, self.epochs,
                                        self.batch_size, None, C.c_uint64(self.seed),
                                        int(self.loop_ids[g.a]), g.epochs_seen,
                                        C.byref(self._adam), None, sp))
            e1.record()
            self._ev.append((e0, e1))
            steps = -(-N // self.batch_size)
            self.stats["fit_bytes"].append(Lg * self.epochs * (4 * N * (D + 1)
                                                               + steps * 24 * self.P))
            g.epochs_seen += self.epochs
            if self.select == "device":     # candidates recomputed in the screening kernel
                _lib.check(lib.bore_sample_screen_topk(
                    C.byref(self.desc), Lg, ptr(th), C.c_uint64(self.seed), int(self.loop_ids[g.a]),
                    g.draws, self.num_samples, self._lo_p, self._hi_p, R, ptr(g.x0), ptr(g.idx),
                    None, sp))
            else:                           # the two-launch statement of the same step
                _lib.check(lib.bore_uniform_candidates(C.c_uint64(self.seed),
                                                       int(self.loop_ids[g.a]), Lg, g.draws,
                                                       self.num_samples, D, self._lo_p, self._hi_p,
                                                       ptr(g.Xc), sp))
                _lib.check(lib.bore_screen_topk(C.byref(self.desc), Lg, ptr(th), ptr(g.Xc),
                                                self.num_samples, 0, R, ptr(g.x0), ptr(g.idx),
                                                None, sp))
            g.draws += 1
            e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e2.record()
            _lib.check(lib.bore_lbfgsb_minimize(C.byref(self.desc), Lg, ptr(th),
                                                _lib.TRANSFORM[tr.name], int(tr.negate),
                                                ptr(g.x0), R, self._lo_p, self._hi_p,
                                                C.byref(self._lopts), ptr(g.x), ptr(g.fun),
                                                ptr(g.jac), ptr(g.info), sp))
            e3.record()
            self._ev2.append((e2, e3))
            if dev_sel:
                st = g.store
                _lib.check(lib.bore_select_best(
                    Lg, R, D, ptr(g.x), ptr(g.fun), ptr(g.info),
                    ptr(st.X) if self.deduplicate else ptr(None), st.n, st.cap, 1e-5, 1e-8,
                    ptr(g.x_best), ptr(g.best), sp))
                g.x_best_pin.copy_(g.x_best, non_blocking=True)
                g.best_pin.copy_(g.best, non_blocking=True)
            else:
                g.x_pin.copy_(g.x, non_blocking=True)
                g.fun_pin.copy_(g.fun, non_blocking=True)
            g.info_pin.copy_(g.info, non_blocking=True)     # (nit, nfev: the bench's byte counts)
            g.done.record()
        g.inflight = True
        self.stats["host_enqueue_s"] = self.stats.get("host_enqueue_s", 0.0) + time.perf_counter() - t_host0

    def _finalize(self, g):
        """Host side of a finished iteration of group g: pick each loop's suggestion, evaluate
        the objective, append."""
        t_host0 = time.perf_counter()
        D = self.D
        info = g.info_pin.numpy()
        nfev = info[:, :, 1]
        self.stats["n_fg_rows"] += int(nfev.sum())
        self.stats["n_rounds"] += int(nfev.max())
        # every f/g row reads x and writes val+grad; every round of a loop
        # re-reads its theta (streaming model)
        self.stats["argmax_bytes"].append(int(nfev.sum()) * 4 * (2 * D + 1)
                                          + int(nfev.max(axis=1).sum()) * 4 * self.P)
        if self.select == "device":
            x_next = g.x_best_pin.numpy().copy()
            none = g.best_pin.numpy() < 0
        else:
            x, fun = g.x_pin.numpy(), g.fun_pin.numpy()
            ok = (info[:, :, 2] == 0) | (info[:, :, 2] == 1)      # success or status == 1
            if self.deduplicate:                                   # Record.is_duplicate as filter_fn
                for l in range(g.b - g.a):
                    for r in range(x.shape[1]):
                        if ok[l, r] and any(np.allclose(xp, x[l, r], rtol=1e-5, atol=1e-8)
                                            for xp in g.X[l]):
                            ok[l, r] = False
            f = np.where(ok, fun, np.inf)
            best = np.argmin(f, axis=1)                            # ties keep the earliest
            x_next = x[np.arange(g.b - g.a), best].copy()
            none = ~ok.any(axis=1)
        if none.any():                                             # reference: random fallback
            self.stats["none_results"] += int(none.sum())
            for l in np.nonzero(none)[0]:
                x_next[l] = self.rs[g.a + l].uniform(self.low, self.high)
        y_next = self.objective(x_next)
        g.X = np.concatenate([g.X, x_next[:, None, :]], axis=1)
        g.y = np.concatenate([g.y, y_next[:, None]], axis=1)
        g.inflight = False
        g.steps += 1
        self.stats["host_finalize_s"] = self.stats.get("host_finalize_s", 0.0) + time.perf_counter() - t_host0
        return x_next, y_next

    def run(self, n_steps):
        """Advance every loop by n_steps BO iterations.  Groups proceed independently; the
        host only reacts to completion events."""
        if self.mode != "device":
            for _ in range(n_steps):
                self.step()
            return
        target = [g.steps + n_steps for g in self.groups]
        for g in self.groups:
            self._enqueue(g)
        remaining = len(self.groups)
        while remaining:
            for i, g in enumerate(self.groups):
                if g.inflight and g.done.query():
                    self._finalize(g)
                    if g.steps < target[i]:
                        self._enqueue(g)
                    else:
                        remaining -= 1

    def step(self):
        """One BO iteration of every loop; returns (x_next [L, D], y_next [L])."""
        if self.mode == "device":
            for g in self.groups:
                self._enqueue(g)
            outs = []
            for g in self.groups:
                g.done.synchronize()
                outs.append(self._finalize(g))
            return (np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs]))
        z = self.label()
        self.fit(z)
        results = self.restarts(self.screen())
        x_next = self.suggest(results)
        y_next = self.objective(x_next)
        g0 = self.groups[0]
        g0.X = np.concatenate([g0.X, x_next[:, None, :]], axis=1)
        g0.y = np.concatenate([g0.y, y_next[:, None]], axis=1)
        return x_next, y_next

    def finish_timing(self):
        torch.cuda.synchronize()
        for e0, e1 in self._ev:
            self.stats["fit_ms"].append(e0.elapsed_time(e1))
        for e0, e1 in self._ev2:
            self.stats["argmax_ms"].append(e0.elapsed_time(e1))
        self._ev, self._ev2 = [], []

    def best(self):
        X, y = self.X, self.y
        i = np.argmin(y, axis=1)
        return X[np.arange(self.L), i], y[np.arange(self.L), i]


class _Group:
    """A contiguous block of loops that steps together on its own stream."""

    def __init__(self, a, b, X, y, eng):
        self.a, self.b = a, b
        self.X, self.y = np.array(X), np.array(y)
        self.epochs_seen = self.draws = self.steps = 0
        self.inflight = False
        if eng.mode == "device":
            R, D = eng.num_starts, eng.D
            self.stream = torch.cuda.Stream(device=eng.device)
            self.done = torch.cuda.Event()
            Lg, dev = b - a, eng.device
            self.x_pin = torch.empty((Lg, R, D), dtype=torch.float64).pin_memory()
            self.fun_pin = torch.empty((Lg, R), dtype=torch.float64).pin_memory()
            self.info_pin = torch.empty((Lg, R, 5), dtype=torch.int32).pin_memory()
            self.Xc = torch.empty((Lg, eng.num_samples, D), dtype=torch.float64, device=dev)
            self.x0 = torch.empty((Lg, R, D), dtype=torch.float64, device=dev)
            self.idx = torch.empty((Lg, R), dtype=torch.int32, device=dev)
            self.x = torch.empty((Lg, R, D), dtype=torch.float64, device=dev)
            self.jac = torch.empty((Lg, R, D), dtype=torch.float64, device=dev)
            self.fun = torch.empty((Lg, R), dtype=torch.float64, device=dev)
            self.info = torch.empty((Lg, R, 5), dtype=torch.int32, device=dev)
            self.cap = 0
            if eng.select == "device":
                n0 = self.X.shape[1]
                self.store = ops.ObservationStore(Lg, D, max(256, 2 * n0), device=dev)
                self.store.load(self.X, self.y)
                self.new_x_pin = torch.empty((Lg, D), dtype=torch.float64).pin_memory()
                self.new_y_pin = torch.empty(Lg, dtype=torch.float64).pin_memory()
                self.new_x_pin_np, self.new_y_pin_np = self.new_x_pin.numpy(), self.new_y_pin.numpy()
                self.new_x = torch.empty((Lg, D), dtype=torch.float64, device=dev)
                self.new_y = torch.empty(Lg, dtype=torch.float64, device=dev)
                self.x_best = torch.empty((Lg, D), dtype=torch.float64, device=dev)
                self.best = torch.empty(Lg, dtype=torch.int32, device=dev)
                self.x_best_pin = torch.empty((Lg, D), dtype=torch.float64).pin_memory()
                self.best_pin = torch.empty(Lg, dtype=torch.int32).pin_memory()
                self.z_dev = torch.empty(Lg * self.store.cap, dtype=torch.float32, device=dev)
            else:
                self.alloc_inputs(eng, max(256, 2 * self.X.shape[1]))

    def alloc_inputs(self, eng, cap):
        """(Re)allocate the observation buffers for up to `cap` points per loop."""
        Lg, D, dev = self.b - self.a, eng.D, eng.device
        torch.cuda.synchronize()
        self.cap = int(cap)
        self.X_pin = torch.empty(Lg * self.cap * D, dtype=torch.float32).pin_memory()
        self.y_pin = torch.empty(Lg * self.cap, dtype=torch.float64).pin_memory()
        self.X_pin_np, self.y_pin_np = self.X_pin.numpy(), self.y_pin.numpy()
        self.X_dev = torch.empty(Lg * self.cap * D, dtype=torch.float32, device=dev)
        self.y_dev = torch.empty(Lg * self.cap, dtype=torch.float64, device=dev)
        self.z_dev = torch.empty(Lg * self.cap, dtype=torch.float32, device=dev)


def shard_loop_ids(rank, world_size, loops_per_gpu):
    """Loop ids owned by `rank`: the contiguous block [rank*L, (rank+1)*L).  Weak scaling: the
    job has world_size*L loops; no loop is shared, none is dropped."""
    return rank * loops_per_gpu + np.arange(loops_per_gpu, dtype=np.int64)


def gather_results(engine, world_size):
    """The only collective of the path: per-loop (id, best x, best y) to rank 0."""
    import torch.distributed as dist
    xb, yb = engine.best()
    mine = torch.from_numpy(np.concatenate([engine.loop_ids[:, None].astype(np.float64), xb,
                                            yb[:, None]], axis=1))
    if world_size == 1:
        return mine.numpy()
    backend = dist.get_backend()
    mine = mine.to(engine.device) if backend == "nccl" else mine
    out = [torch.empty_like(mine) for _ in range(world_size)] if dist.get_rank() == 0 else None
    dist.gather(mine, out, dst=0)
    if dist.get_rank() == 0:
        allr = torch.cat(out).cpu().numpy()
        return allr[np.argsort(allr[:, 0])]
    return None
