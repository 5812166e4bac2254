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


class ReplicaEngine:
    def __init__(self, loop_ids, input_dim=2, units=(16, 16, 1), acts=("relu", "relu", "sigmoid"),
                 transform="identity", gamma=0.25, epochs=200, batch_size=64, num_starts=3,
                 num_samples=1024, n_init=10, objective=branin01, max_points=None,
                 options=None, device=None, mode="device", seed=0):
        """mode "device": label, fit, candidate draw, screening and all L-BFGS-B restarts run
        as five launches per BO iteration with ONE host sync (candidates from the device
        counter stream).  mode "lockstep": candidates from each loop's numpy RandomState and
        SciPy's L-BFGS-B on the host around the batched f/g kernel (the reference's streams
        and optimiser code; slow -- one launch + L*R host state machines per round)."""
        assert mode in ("device", "lockstep")
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
        self.gamma, self.epochs, self.batch_size = gamma, int(epochs), int(batch_size)
        self.num_starts, self.num_samples = int(num_starts), int(num_samples)
        self.objective = objective
        self.options = dict(options or dict(maxiter=1000, ftol=1e-9))
        self.low, self.high = np.zeros(D), np.ones(D)
        # per-loop host RNG: the stream the reference would consume (RandomState(seed))
        self.rs = [np.random.RandomState(int(s)) for s in self.loop_ids]
        # Keras-default initial weights, one model per loop
        th = np.empty((L, self.P), dtype=np.float32)
        for i, rs in enumerate(self.rs):
            off, fan_in = 0, D
            for u in self.units:
                lim = np.sqrt(6.0 / (fan_in + u))
                th[i, off:off + fan_in * u] = rs.uniform(-lim, lim, size=fan_in * u)
                off += fan_in * u
                th[i, off:off + u] = 0.0
                off += u
                fan_in = u
        self.theta = torch.from_numpy(th).to(self.device)
        self.adam_m = torch.zeros_like(self.theta)
        self.adam_v = torch.zeros_like(self.theta)
        self.adam_t = torch.zeros(L, dtype=torch.int64, device=self.device)
        self.epochs_seen = 0
        self.draws = 0
        # observations
        self.X = np.stack([rs.uniform(self.low, self.high, size=(n_init, D)) for rs in self.rs])
        self.y = self.objective(self.X)
        self.stats = dict(fit_ms=[], fit_bytes=[], argmax_ms=[], argmax_bytes=[], n_fg_rows=0,
                          n_rounds=0, none_results=0)
        self._ev, self._ev2 = [], []

    @property
    def N(self):
        return self.X.shape[1]

    # -- stages ------------------------------------------------------------
    def label(self):
        tau = np.quantile(self.y, q=self.gamma, axis=1)
        return np.less(self.y, tau[:, None])

    def fit(self, z):
        L, N, D = self.L, self.N, self.D
        Xd = torch.from_numpy(self.X.astype(np.float32)).to(self.device)
        zd = torch.from_numpy(z.astype(np.float32)).to(self.device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mlp_fit(self.desc, self.theta, self.adam_m, self.adam_v, self.adam_t, Xd, zd,
                    self.epochs, self.batch_size, seed=self.seed,
                    model_index0=int(self.loop_ids[0]), epoch0=self.epochs_seen, want_loss=False)
        e1.record()
        self._ev.append((e0, e1))
        steps = -(-N // self.batch_size)
        self.stats["fit_bytes"].append(L * self.epochs * (4 * N * (D + 1) + steps * 24 * self.P))
        self.epochs_seen += self.epochs

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

    def step_device(self):
        """One BO iteration of every loop, device-resident: 5 launches, 1 sync."""
        L, N, D, R = self.L, self.N, self.D, self.num_starts
        dev = self.device
        Xd = torch.from_numpy(self.X.astype(np.float32)).to(dev, non_blocking=True)
        yd = torch.from_numpy(self.y).to(dev, non_blocking=True)
        zd = ops.labels(yd, self.gamma)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.mlp_fit(self.desc, self.theta, self.adam_m, self.adam_v, self.adam_t, Xd, zd,
                    self.epochs, self.batch_size, seed=self.seed,
                    model_index0=int(self.loop_ids[0]), epoch0=self.epochs_seen, want_loss=False)
        e1.record()
        self._ev.append((e0, e1))
        steps = -(-N // self.batch_size)
        self.stats["fit_bytes"].append(L * self.epochs * (4 * N * (D + 1) + steps * 24 * self.P))
        self.epochs_seen += self.epochs
        Xc = ops.uniform_candidates(self.seed, L, self.num_samples, self.low, self.high,
                                    model_index0=int(self.loop_ids[0]), draw_index=self.draws,
                                    device=dev)
        self.draws += 1
        x0, _ = ops.screen_topk(self.desc, self.theta, Xc, R)
        tr = self.transform.negated()
        e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e2.record()
        x, fun, _, info = ops.lbfgsb_minimize(self.desc, self.theta, x0, self.low, self.high,
                                              tr.name, tr.negate, **self.options)
        e3.record()
        self._ev2.append((e2, e3))
        x, fun, info = x.cpu().numpy(), fun.cpu().numpy(), info.cpu().numpy()   # the one sync
        nfev = info[:, :, 1]
        self.stats["n_fg_rows"] += int(nfev.sum())
        self.stats["n_rounds"] += int(nfev.max())
        # every f/g row reads x and writes val+grad; every round of a loop
        # re-reads its theta (streaming model)
        self.stats["argmax_bytes"].append(int(nfev.sum()) * 4 * (2 * D + 1)
                                          + int(nfev.max(axis=1).sum()) * 4 * self.P)
        ok = (info[:, :, 2] == 0) | (info[:, :, 2] == 1)          # success or status == 1
        f = np.where(ok, fun, np.inf)
        best = np.argmin(f, axis=1)                                # ties keep the earliest
        x_next = x[np.arange(L), best]
        none = ~ok.any(axis=1)
        if none.any():                                             # reference: random fallback
            self.stats["none_results"] += int(none.sum())
            for l in np.nonzero(none)[0]:
                x_next[l] = self.rs[l].uniform(self.low, self.high)
        return x_next

    def step(self):
        """One BO iteration of every loop."""
        if self.mode == "device":
            x_next = self.step_device()
        else:
            z = self.label()
            self.fit(z)
            results = self.restarts(self.screen())
            x_next = self.suggest(results)
        y_next = self.objective(x_next)
        self.X = np.concatenate([self.X, x_next[:, None, :]], axis=1)
        self.y = np.concatenate([self.y, y_next[:, None]], axis=1)
        return x_next, y_next

    def finish_timing(self):
        torch.cuda.synchronize()
        for e0, e1 in self._ev:
            self.stats["fit_ms"].append(e0.elapsed_time(e1))
        for e0, e1 in self._ev2:
            self.stats["argmax_ms"].append(e0.elapsed_time(e1))
        self._ev, self._ev2 = [], []

    def best(self):
        i = np.argmin(self.y, axis=1)
        return self.X[np.arange(self.L), i], self.y[np.arange(self.L), i]


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
