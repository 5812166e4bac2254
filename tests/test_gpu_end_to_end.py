"""The kernel bench.py times, end to end against the oracle.

`iteration_kernel` (bore_amd/csrc/bore_iter.hip: append -> labels -> fit -> sample + screen ->
L-BFGS-B restarts -> pick, one launch per loop-iteration, asynchronous schedule) is compared with
the reference's loop (README.rst:83-103, bore/mixins.py:22-89) restated by the oracle, BO
iteration by BO iteration, on the same streams: the epoch shuffles (bore_amd.shuffle) and the
candidate draws (bore_amd.sampling) are counter streams with host statements, the initial
weights / observations and the None fall-back come from numpy's RandomState(loop id).

Two forms:
  * teacher-forced -- every iteration starts from the ENGINE's state and record, so each fit and
    each argmax is checked on its own (no compounding): theta after the fit against oracle.fit,
    the suggestion against oracle.argmax (scipy L-BFGS-B, sequential restarts) on the fitted
    network;
  * free-running  -- the oracle runs its own loop from the same start; the rate of loops whose
    whole trajectory stays with the engine's is reported and bounded from below.

Stated tolerances: theta after one fit (200-400 fp32 Adam steps from the same state) within
2e-4 + 2e-3*|ref| (median error is ~1e-6; a relu unit switching on one row moves single weights
more).  The argmax is compared restart by restart (the engine's suggestion is first shown to be,
bit for bit, the pick of the separately tested screen / L-BFGS-B / pick kernels, whose
per-restart results are visible): with an fp32 objective and ftol 1e-9 whether a restart ends
"converged" or "abnormal termination in the line search" hangs on the last bits of f, and the
oracle's numpy arithmetic and the kernel's MFMA chains differ in exactly those (so would
TensorFlow's) -- the reference's acceptance rule (bore/mixins.py:80-89) then picks another restart.
Bounded from below (measured values beside the asserts): the screening picks the same starts
(>= 90 %), the acceptance of a restart agrees (>= 75 %), accepted restarts end within 1e-5 -- the
reference's own duplicate tolerance, bore/data.py:43-48 -- of scipy's (>= 93 %), and where all
acceptances agree the suggestion is the oracle's (>= 90 %).  tests/test_gpu_agreement.py removes the arithmetic difference (scipy on
the kernel's own f/g) and finds the device optimiser's pick equal to scipy's.
"""
import numpy as np
import pytest
from scipy.optimize import Bounds

from bore_amd import sampling, shuffle
from oracle import bore_oracle as O
from test_gpu_parity import pack, unpack
from conftest import record_measurement

pytestmark = pytest.mark.gpu

D, UNITS, ACTS = 2, [16, 16, 1], ["relu", "relu", "sigmoid"]
LOW, HIGH = np.zeros(D), np.ones(D)
BOUNDS = Bounds(LOW, HIGH)


def _oracle_iteration(p, st, X, y, loop_id, it, rs, epochs=200, num_samples=1024, num_starts=3):
    """One BO iteration of the reference's loop on the oracle: returns (suggestion, was_none).
    p / st are updated in place by the fit."""
    z, _ = O.labels(y, 0.25)
    N = len(y)
    perms = shuffle.permutations(0, 1, epochs, N, model_index0=loop_id, epoch0=it * epochs)[0]
    O.fit(p, ACTS, st, X, z, perms, batch_size=64)
    Xc = sampling.uniform_candidates(0, 1, num_samples, LOW, HIGH, model_index0=loop_id,
                                     draw_index=it)[0]
    res = O.argmax(p, ACTS, BOUNDS, num_starts=num_starts, num_samples=num_samples, X_init=Xc)
    if res is None:
        return rs.uniform(LOW, HIGH), True, None
    return res.x, False, res


def _adam_state(p, m, v, t):
    st = O.AdamState(p)
    st.m, st.v, st.t = unpack(m.copy(), D, UNITS), unpack(v.copy(), D, UNITS), int(t)
    return st


def test_fused_iteration_kernel_against_the_oracle_teacher_forced(gpu):
    import torch
    from bore_amd import _lib, ops
    from bore_amd.engine import NativeEngine, branin01, initial_state
    ids = np.arange(40, 56)
    L, T, R = len(ids), 5, 3
    eng = NativeEngine(ids, async_loops=True)          # BASELINE config 1 defaults
    P = eng.P
    desc = _lib.make_desc(D, UNITS, ACTS)
    rss, th0, X0, y0 = initial_state(ids, D, UNITS, P, 10, branin01, LOW, HIGH)
    assert np.array_equal(eng.state()[0], th0) and np.array_equal(eng.X, X0)
    worst_theta = 0.0
    n_iter = n_same_starts = n_pick_same = n_clean = n_clean_same = n_none_both = n_none_one = 0
    n_rest = n_rest_accept_agree = n_rest_both = n_rest_both_same = 0
    for it in range(T):
        th_prev, m_prev, v_prev, t_prev = eng.state()
        X_prev, y_prev = eng.observations()
        eng.run(1)
        th, m, v, t = eng.state()
        X_new, y_new = eng.observations()
        N = X_prev.shape[1]
        assert X_new.shape[1] == N + 1 and np.array_equal(X_new[:, :N], X_prev)
        assert np.array_equal(t, t_prev + 200 * -(-N // 64))
        # the same argmax through the separate kernels (screen, restarts, pick): per-restart
        # results of what the fused kernel did (bit-identical -- asserted on the pick below)
        thd = torch.from_numpy(th).cuda()
        x0d, _ = ops.sample_screen_topk(desc, thd, 0, 1024, LOW, HIGH, R, model_index0=int(ids[0]),
                                        draw_index=it)
        xd, fund, _, infod = ops.lbfgsb_minimize(desc, thd, x0d, LOW, HIGH, "identity", True,
                                                 maxiter=1000, ftol=1e-9)
        xbd, bestd = ops.select_best(xd, fund, infod)
        x0d, xd, fund, infod, xbd, bestd = (a.cpu().numpy() for a in (x0d, xd, fund, infod, xbd, bestd))
        for l in range(L):
            # -- the fit, from the engine's previous state, on the engine's record
            p = unpack(th_prev[l].copy(), D, UNITS)
            st = _adam_state(p, m_prev[l], v_prev[l], t_prev[l])
            z, _ = O.labels(y_prev[l], 0.25)
            perms = shuffle.permutations(0, 1, 200, N, model_index0=int(ids[l]), epoch0=it * 200)[0]
            O.fit(p, ACTS, st, X_prev[l], z, perms, batch_size=64)
            ref = pack(p)
            err = np.abs(th[l] - ref) / (2e-4 + 2e-3 * np.abs(ref))
            worst_theta = max(worst_theta, float(err.max()))
            assert err.max() <= 1.0, (it, l, float(err.max()))
            assert st.t == t[l]
            # -- the engine's suggestion is the pick of the separate kernels, bit for bit
            x_eng = X_new[l, N]
            assert y_new[l, N] == branin01(x_eng)
            if bestd[l] >= 0:
                assert np.array_equal(x_eng, xbd[l])
            # -- the argmax of the oracle on the network the ENGINE fitted
            pe = unpack(th[l].copy(), D, UNITS)
            Xc = sampling.uniform_candidates(0, 1, 1024, LOW, HIGH, model_index0=int(ids[l]),
                                             draw_index=it)[0]
            pred = O.predict(pe, ACTS, Xc).squeeze(axis=-1)
            starts = Xc[np.argpartition(-pred, kth=R - 1, axis=None)[:R]]     # bore/mixins.py:53-57
            results = O.maxima(pe, ACTS, BOUNDS, num_starts=R, num_samples=1024, X_init=Xc)
            n_iter += 1
            same_starts = ({tuple(r) for r in starts} == {tuple(r) for r in x0d[l]})
            n_same_starts += same_starts
            best = None
            for res in results:
                if (res.success or res.status == 1) and (best is None or res.fun < best.fun):
                    best = res
            if best is None or bestd[l] < 0:
                n_none_both += (best is None) and bestd[l] < 0
                n_none_one += (best is None) != (bestd[l] < 0)
            pick_same = (best is not None and bestd[l] >= 0
                         and np.allclose(x_eng, best.x, rtol=0, atol=1e-5))
            n_pick_same += pick_same
            if not same_starts:
                continue
            clean = True
            for r in range(R):              # restart by restart, matched by start point
                k = [tuple(s) for s in starts].index(tuple(x0d[l, r]))
                res = results[k]
                acc_o, acc_d = bool(res.success or res.status == 1), infod[l, r, 2] in (0, 1)
                n_rest += 1
                n_rest_accept_agree += acc_o == acc_d
                clean = clean and acc_o == acc_d
                if acc_o and acc_d:
                    n_rest_both += 1
                    n_rest_both_same += np.allclose(xd[l, r], res.x, rtol=0, atol=1e-5)
            if clean and best is not None and bestd[l] >= 0:
                n_clean += 1
                n_clean_same += pick_same
    print(f"\n[end-to-end, teacher-forced] {L} loops x {T} iterations; worst theta error / tolerance "
          f"{worst_theta:.3f}; screening picks the same starts in {n_same_starts}/{n_iter}; restarts: "
          f"acceptance (success or status 1) agrees with scipy-on-the-oracle in "
          f"{n_rest_accept_agree}/{n_rest}, both accepted and x within 1e-5 in {n_rest_both_same}/"
          f"{n_rest_both}; suggestion within 1e-5 of the oracle's: {n_pick_same}/{n_iter} overall, "
          f"{n_clean_same}/{n_clean} where every restart's acceptance agrees; None on both sides "
          f"{n_none_both}, on one side {n_none_one}")
    record_measurement("end_to_end_teacher_forced_cfg1", dict(
        loops=L, iterations=T, worst_theta_error_over_tolerance=worst_theta,
        same_starts=[int(n_same_starts), n_iter], acceptance_agrees=[int(n_rest_accept_agree), n_rest],
        accepted_within_1e5=[int(n_rest_both_same), n_rest_both],
        pick_same_overall=[int(n_pick_same), n_iter], pick_same_clean=[int(n_clean_same), n_clean],
        none_both=int(n_none_both), none_one_side=int(n_none_one)))
    # measured (r2, 16 loops x 5 iterations): same starts 77/80 (0.96); acceptance agrees 192/231
    # (0.83); accepted restarts within 1e-5: 166/172 (0.97); suggestion 48/50 (0.96) where all
    # acceptances agree, 64/80 (0.80) overall.  Floors = measured - 5 points.
    assert n_same_starts >= 0.93 * n_iter
    assert n_rest_accept_agree >= 0.78 * n_rest
    assert n_rest_both_same >= 0.93 * n_rest_both
    assert n_clean_same >= 0.92 * n_clean and n_clean >= 0.5 * n_iter
    assert n_pick_same >= 0.75 * n_iter


def test_fused_iteration_kernel_against_the_oracle_free_running(gpu):
    from bore_amd.engine import NativeEngine, branin01, initial_state
    ids = np.arange(7, 15)
    L, T = len(ids), 4
    eng = NativeEngine(ids, async_loops=True)
    rss, th0, X0, y0 = initial_state(ids, D, UNITS, eng.P, 10, branin01, LOW, HIGH)
    eng.run(T)
    Xe, ye = eng.observations()
    same_loops, first_split = 0, []
    for l in range(L):
        p = unpack(th0[l].copy(), D, UNITS)
        st = O.AdamState(p)
        X, y = X0[l].copy(), y0[l].copy()
        ok = True
        for it in range(T):
            x, _, _ = _oracle_iteration(p, st, X, y, int(ids[l]), it, rss[l])
            if not np.allclose(x, Xe[l, 10 + it], rtol=0, atol=1e-4):
                ok = False
                first_split.append(it)
                break
            X, y = np.vstack([X, x]), np.append(y, branin01(x))
        same_loops += ok
    print(f"\n[end-to-end, free-running] {same_loops}/{L} loops keep the oracle's trajectory (1e-4) "
          f"for {T} iterations; first differing iteration of the others: {first_split}")
    record_measurement("end_to_end_free_running_cfg1", dict(loops=L, iterations=T, same_loops=int(same_loops),
                                                            first_split=[int(i) for i in first_split]))
    # measured (r2): 5/8.  (fp32 noise decides line searches: see the module docstring)
    assert same_loops >= L // 2


def test_config2_fit_and_argmax_chain_against_the_oracle_teacher_forced(gpu):
    """One wide BASELINE config end to end: config 2 (Hartmann-6 box, 6 ->
    32-32-1, many restarts) through the SAME launch chain the replica engine and bench.py's
    `configs` leg use for wide models -- fit_kernel<2> -> sample + screen -> lbfgsb_kernel<2> ->
    pick -- against oracle.fit + oracle.maxima / the reference's acceptance rule
    (bore/mixins.py:57-89), iteration by iteration, teacher-forced (every iteration starts from
    the kernels' state; the record grows by the kernels' suggestion)."""
    import torch
    from bore_amd import _lib, ops
    D2, U2, A2 = 6, [32, 32, 1], ["relu", "relu", "sigmoid"]
    lo, hi = np.zeros(D2), np.ones(D2)
    bounds = Bounds(lo, hi)
    L, T, R, NS, E, N0 = 4, 3, 32, 1024, 100, 96
    desc = _lib.make_desc(D2, U2, A2)
    P = ops.param_count(desc)
    rs = np.random.RandomState(21)
    th0 = np.stack([pack(O.glorot_uniform_params(D2, U2, rs)) for _ in range(L)])
    c = rs.uniform(0.2, 0.8, size=D2)

    def objective(X):
        return np.sum((X - c) ** 2, axis=-1) + 0.1 * np.sin(5.0 * X.sum(axis=-1))

    X = rs.uniform(size=(L, N0, D2))
    y = objective(X)
    theta = torch.from_numpy(th0).cuda()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    t = torch.zeros(L, dtype=torch.int64, device=theta.device)
    worst_theta = 0.0
    n_iter = n_same_starts = n_pick_same = n_clean = n_clean_same = 0
    n_rest = n_acc_agree = n_both = n_both_same = 0
    dfun, regret = [], []
    for it in range(T):
        N = X.shape[1]
        z = np.stack([O.labels(y[l], 0.25)[0] for l in range(L)]).astype(np.float32)
        th_prev, m_prev, v_prev, t_prev = (a.cpu().numpy().copy() for a in (theta, m, v, t))
        ops.mlp_fit(desc, theta, m, v, t, torch.from_numpy(X.astype(np.float32)).cuda(),
                    torch.from_numpy(z).cuda(), E, 64, seed=5, model_index0=100, epoch0=it * E,
                    want_loss=False)
        x0d, _ = ops.sample_screen_topk(desc, theta, 5, NS, lo, hi, R, model_index0=100, draw_index=it)
        xd, fund, _, infod = ops.lbfgsb_minimize(desc, theta, x0d, lo, hi, "identity", True,
                                                 maxiter=1000, ftol=1e-9)
        xbd, bestd = ops.select_best(xd, fund, infod)
        th = theta.cpu().numpy()
        x0d, xd, fund, infod, xbd, bestd = (a.cpu().numpy() for a in (x0d, xd, fund, infod, xbd, bestd))
        assert np.array_equal(t.cpu().numpy(), t_prev + E * -(-N // 64))
        x_next = np.empty((L, D2))
        for l in range(L):
            p = unpack(th_prev[l].copy(), D2, U2)
            st = O.AdamState(p)
            st.m, st.v, st.t = unpack(m_prev[l].copy(), D2, U2), unpack(v_prev[l].copy(), D2, U2), int(t_prev[l])
            perms = shuffle.permutations(5, 1, E, N, model_index0=100 + l, epoch0=it * E)[0]
            O.fit(p, A2, st, X[l], z[l], perms, batch_size=64)
            ref = pack(p)
            err = np.abs(th[l] - ref) / (2e-4 + 2e-3 * np.abs(ref))
            worst_theta = max(worst_theta, float(err.max()))
            assert err.max() <= 1.0, (it, l, float(err.max()))
            # the oracle's argmax on the network the kernels fitted
            pe = unpack(th[l].copy(), D2, U2)
            Xc = sampling.uniform_candidates(5, 1, NS, lo, hi, model_index0=100 + l, draw_index=it)[0]
            pred = O.predict(pe, A2, Xc).squeeze(axis=-1)
            starts = Xc[np.argpartition(-pred, kth=R - 1, axis=None)[:R]]
            results = O.maxima(pe, A2, bounds, num_starts=R, num_samples=NS, X_init=Xc)
            n_iter += 1
            same_starts = ({tuple(r) for r in starts} == {tuple(r) for r in x0d[l]})
            n_same_starts += same_starts
            best = None
            for res in results:
                if (res.success or res.status == 1) and (best is None or res.fun < best.fun):
                    best = res
            x_eng = xbd[l] if bestd[l] >= 0 else rs.uniform(lo, hi)
            x_next[l] = x_eng
            # "same pick": the same point, or an equally good one (a saturated sigmoid is flat:
            # several restarts end at the same value in different corners)
            pick_same = (best is not None and bestd[l] >= 0
                         and (np.allclose(x_eng, best.x, rtol=0, atol=1e-5)
                              or abs(fund[l, bestd[l]] - best.fun) <= 2e-6))
            n_pick_same += pick_same
            if best is not None and bestd[l] >= 0:
                # the kernels' suggestion under the ORACLE's objective against the oracle's own best
                v_dev = float(O.value_and_input_grad(pe, A2, x_eng[None, :], "identity")[0][0])
                regret.append(v_dev - float(best.fun))
            if not same_starts:
                continue
            clean = True
            for r in range(R):
                k = [tuple(s) for s in starts].index(tuple(x0d[l, r]))
                res = results[k]
                acc_o, acc_d = bool(res.success or res.status == 1), infod[l, r, 2] in (0, 1)
                n_rest += 1
                n_acc_agree += acc_o == acc_d
                clean = clean and acc_o == acc_d
                dfun.append(float(fund[l, r]) - float(res.fun))
                if acc_o and acc_d:
                    n_both += 1
                    n_both_same += np.allclose(xd[l, r], res.x, rtol=0, atol=1e-5)
            if clean and best is not None and bestd[l] >= 0:
                n_clean += 1
                n_clean_same += pick_same
        X = np.concatenate([X, x_next[:, None, :]], axis=1)
        y = np.concatenate([y, objective(x_next)[:, None]], axis=1)
    print(f"\n[end-to-end, config 2, teacher-forced] {L} models x {T} iterations x {R} restarts; worst "
          f"theta error / tolerance {worst_theta:.3f}; same starts {n_same_starts}/{n_iter}; acceptance "
          f"agrees {n_acc_agree}/{n_rest}; both accepted and x within 1e-5: {n_both_same}/{n_both}; pick "
          f"equal or equally good: {n_pick_same}/{n_iter} overall, {n_clean_same}/{n_clean} where every "
          f"restart's acceptance agrees")
    dfun, regret = np.abs(dfun), np.asarray(regret)
    print("|dfun| quantiles 50/90/99/max:", np.quantile(dfun, [0.5, 0.9, 0.99, 1.0]), "regret:", np.sort(regret))
    record_measurement("end_to_end_teacher_forced_cfg2", dict(
        abs_dfun_q50_q90_q99_max=[float(q) for q in np.quantile(dfun, [0.5, 0.9, 0.99, 1.0])],
        pick_regret_under_oracle_objective=[float(q) for q in np.sort(regret)],
        models=L, iterations=T, restarts=R, worst_theta_error_over_tolerance=worst_theta,
        same_starts=[int(n_same_starts), n_iter], acceptance_agrees=[int(n_acc_agree), n_rest],
        accepted_within_1e5=[int(n_both_same), n_both], pick_same_overall=[int(n_pick_same), n_iter],
        pick_same_clean=[int(n_clean_same), n_clean]))
    # Measured (r3, profiles/r3/parity_measured.json): same starts 12/12; acceptance agrees 270/384
    # (0.70); accepted restarts within 1e-5 in x: 64/215 (0.30) -- in six dimensions, on a barely
    # trained classifier, two float32 statements of the SAME objective (numpy's BLAS sums vs the
    # kernel's k-ordered fmaf chains) send most L-BFGS-B runs to different points of equal quality:
    # |fun - fun_oracle| median 5e-7, 90 % within 7e-4 -- and the device optimiser fed the kernel's
    # own f/g is scipy's (tests/test_gpu_agreement.py: 0.995 of restarts identical).  What the
    # reference's caller consumes is the pick: under the ORACLE's objective the kernels' suggestion
    # is within 2.4e-4 of the oracle's own best in 12/12 iterations (better in 3), within 5e-6 in 9/12.
    assert n_same_starts >= 0.9 * n_iter
    assert n_acc_agree >= 0.62 * n_rest
    assert n_both_same >= 0.22 * n_both
    assert np.median(dfun) <= 5e-6 and np.quantile(dfun, 0.9) <= 5e-3
    assert len(regret) >= 0.9 * n_iter
    assert np.all(np.abs(regret) <= 1e-3) and np.mean(np.abs(regret) <= 1e-5) >= 0.6
