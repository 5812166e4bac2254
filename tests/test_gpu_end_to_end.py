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
    regret64 = []
    # iterations where the two sides part visibly -- None on one side only, or a pick more than 1e-3 apart under
    # the float64 network -- and what SciPy's own L-BFGS-B makes of the DEVICE's f / g from the device's starts
    odd = dict(none_one_side=0, regret_above_1e3=0, explained_by_float32_objective=0, unexplained=[])
    from bore_amd.optimizers import lockstep
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
            is_odd = bool((best is None) != (bestd[l] < 0))
            odd["none_one_side"] += int(is_odd)
            if best is not None and bestd[l] >= 0:
                # the suggestion valued by the FLOAT64 network against the oracle's own suggestion
                v64 = lambda xx: float(O.value_and_input_grad(pe, ACTS, np.asarray(xx)[None, :], "identity",
                                                              dtype=np.float64)[0][0])
                regret64.append(v64(x_eng) - v64(best.x))
                if abs(regret64[-1]) > 1e-3:
                    is_odd = True
                    odd["regret_above_1e3"] += 1
            if is_odd:
                # Cause: the device's restarts ARE SciPy's restarts on the device's own
                # float32 f / g -- same acceptance, accepted points within 1e-5 -- so what parts the two sides
                # is the objective's arithmetic (float32 kernel against float32 numpy: other summation orders,
                # other last bits, a line search at the noise floor ending one evaluation apart), not the
                # optimiser's logic.
                def fg_dev(Xb, l=l):
                    Xb = np.atleast_2d(Xb)
                    vv, gg = ops.mlp_value_and_input_grad(desc, thd[l:l + 1], torch.from_numpy(Xb[None]).cuda(),
                                                          "identity", True)
                    return vv.cpu().numpy()[0], gg.cpu().numpy()[0]
                ref = lockstep.minimize_lockstep(fg_dev, x0d[l], bounds=BOUNDS, maxiter=1000, ftol=1e-9)
                same = True
                for r in range(R):
                    acc_s, acc_d = bool(ref[r].success or ref[r].status == 1), infod[l, r, 2] in (0, 1)
                    same = same and acc_s == acc_d and (not acc_s or np.allclose(ref[r].x, xd[l, r], rtol=0, atol=1e-5))
                if same:
                    odd["explained_by_float32_objective"] += 1
                else:
                    odd["unexplained"].append([int(it), int(l)])
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
        none_both=int(n_none_both), none_one_side=int(n_none_one), visible_differences_by_cause=odd,
        pick_regret_under_float64_objective_quantiles_0_50_90_100=[
            float(q) for q in np.quantile(np.abs(regret64), [0.0, 0.5, 0.9, 1.0])]))
    # the stated float32 tolerance for suggested candidates: valued by the float64 network, the engine's
    # suggestion is within 1e-3 of the oracle's in at least 95 % of the iterations where both suggest one
    # (a restart that ends on another float32 plateau of the sigmoid accounts for the rest)
    assert np.mean(np.abs(regret64) <= 1e-3) >= 0.95, np.sort(np.abs(regret64))[-5:]
    # ... and every visible difference has the cause named above: SciPy on the device's f / g does what the device did
    print("[end-to-end, teacher-forced] visible differences:", odd)
    assert not odd["unexplained"], odd
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
    dfun, regret, regret64 = [], [], []
    status_pairs, acc_split, other_point = {}, {}, {}
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
                # ... and under the FLOAT64 network: the stated tolerance for suggested candidates
                v64 = lambda xx: float(O.value_and_input_grad(pe, A2, np.asarray(xx)[None, :], "identity",
                                                              dtype=np.float64)[0][0])
                regret64.append(v64(x_eng) - v64(best.x))
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
                # -- per cause --
                key = f"{int(res.status)}/{int(infod[l, r, 2])}"          # scipy-on-the-oracle / device
                status_pairs[key] = status_pairs.get(key, 0) + 1
                if acc_o and acc_d:
                    n_both += 1
                    same_x = bool(np.allclose(xd[l, r], res.x, rtol=0, atol=1e-5))
                    n_both_same += same_x
                    if not same_x:
                        # another point: of equal value (the flat part of a saturated sigmoid, or the
                        # same basin left at another float32 plateau) or a different optimum?
                        d = abs(float(fund[l, r]) - float(res.fun))
                        cause = ("equal_value_2e-6" if d <= 2e-6 else "value_within_1e-3" if d <= 1e-3
                                 else "different_optimum")
                        other_point[cause] = other_point.get(cause, 0) + 1
                elif acc_o != acc_d:
                    # one side ends "abnormal termination in the line search" (status 2): on the last
                    # bits of f (values agree to 2e-6) or at different values?
                    d = abs(float(fund[l, r]) - float(res.fun))
                    cause = "status2_vs_accepted_same_value_2e-6" if d <= 2e-6 else "status2_vs_accepted_other_value"
                    acc_split[cause] = acc_split.get(cause, 0) + 1
            if clean and best is not None and bestd[l] >= 0:
                n_clean += 1
                n_clean_same += pick_same
        X = np.concatenate([X, x_next[:, None, :]], axis=1)
        y = np.concatenate([y, objective(x_next)[:, None]], axis=1)
    print("[config 2] status pairs (oracle/device):", status_pairs, "| acceptance disagreements:", acc_split,
          "| both accepted, other point:", other_point)
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
        pick_regret_under_float64_objective=[float(q) for q in np.sort(regret64)],
        status_pairs_oracle_device=status_pairs, acceptance_disagreements_by_cause=acc_split,
        both_accepted_other_point_by_cause=other_point,
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
    # The decomposition (round 4, recorded): every restart both sides accept but end at different points,
    # and every acceptance disagreement, is counted under exactly one cause ...
    assert sum(other_point.values()) == n_both - n_both_same
    assert sum(acc_split.values()) == n_rest - n_acc_agree
    # ... other points are mostly points of (nearly) EQUAL VALUE on a flat classifier, not other optima
    # (measured r4: 39 within 2e-6, 85 within 1e-3, 15 beyond, of 139), and most acceptance disagreements
    # are a line search that fails on one side at the SAME value (72 of 123 within 2e-6),
    assert other_point.get("different_optimum", 0) <= 0.2 * max(1, n_both - n_both_same)
    assert acc_split.get("status2_vs_accepted_same_value_2e-6", 0) >= 0.4 * max(1, n_rest - n_acc_agree)
    # and valued by the float64 network the suggestion is within 1e-3 of the oracle's (the stated
    # float32 tolerance for suggested candidates)
    assert np.all(np.abs(regret64) <= 1e-3)


def _wide_chain_teacher_forced(name, D2, U2, A2, compute, L, T, R, NS, E, N0, seed, transform="identity", gamma=0.25,
                               l2=None):
    """fit -> sample + screen -> restarts -> pick through the launch chain of a wide model, iteration
    by iteration against the oracle (float32 or mixed-bfloat16 statement), teacher-forced: every
    iteration starts from the kernels' state; the record grows by the kernels' suggestion.  Returns
    the measurements (also recorded into parity_measured.json)."""
    import torch
    from bore_amd import _lib, ops
    bf = compute == "bfloat16"
    lo, hi = np.zeros(D2), np.ones(D2)
    bounds = Bounds(lo, hi)
    # (the plugin's l2_factor: kernel and bias of every HIDDEN layer -- bore/models.py:21-27 passes the
    # regularisers to the hidden Dense layers, the output layer has none)
    l2k = [l2] * (len(U2) - 1) + [0.0] if l2 else None
    desc = _lib.make_desc(D2, U2, A2, l2_kernel=l2k, l2_bias=l2k, compute=compute)
    l2_oracle = [l2] * (2 * len(U2) - 2) + [0.0, 0.0] if l2 else None
    rs = np.random.RandomState(seed)
    th0 = np.stack([pack(O.glorot_uniform_params(D2, U2, rs)) for _ in range(L)])
    c = rs.uniform(0.2, 0.8, size=D2)

    def objective(X):
        return np.sum((X - c) ** 2, axis=-1) + 0.1 * np.sin(5.0 * X.sum(axis=-1))

    def oracle_value64(pe, x):
        """T(-f(x)) of the network in FLOAT64 arithmetic: the yardstick for "how good is this
        suggestion", free of either side's float32 / bfloat16 noise."""
        return float(O.value_and_input_grad(pe, A2, x[None, :], transform, dtype=np.float64)[0][0])

    X = rs.uniform(size=(L, N0, D2))
    y = objective(X)
    theta = torch.from_numpy(th0).cuda()
    m, v = torch.zeros_like(theta), torch.zeros_like(theta)
    t = torch.zeros(L, dtype=torch.int64, device=theta.device)
    worst_theta = 0.0
    n_iter = n_same_starts = n_pick_same = n_none_dev = n_none_ora = 0
    n_rest = n_acc_agree = n_both = n_both_same = 0
    start_overlap, dfun, regret, regret64, theta_within, fit_loss_rel = [], [], [], [], [], []
    bf16_logit_ulps, bf16_causes = [], dict(gate=0, step=0, other=0, weights=0)
    # why an accepted-by-one restart is not accepted by the other: (oracle status, device status)
    status_pairs = {}
    for it in range(T):
        N = X.shape[1]
        z = np.stack([O.labels(y[l], gamma)[0] for l in range(L)]).astype(np.float32)
        th_prev, m_prev, v_prev, t_prev = (a.cpu().numpy().copy() for a in (theta, m, v, t))
        ops.mlp_fit(desc, theta, m, v, t, torch.from_numpy(X.astype(np.float32)).cuda(),
                    torch.from_numpy(z).cuda(), E, 64, seed=5, model_index0=100, epoch0=it * E,
                    want_loss=False)
        x0d, _ = ops.sample_screen_topk(desc, theta, 5, NS, lo, hi, R, model_index0=100, draw_index=it)
        xd, fund, _, infod = ops.lbfgsb_minimize(desc, theta, x0d, lo, hi, transform, True,
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
            if bf:
                O.fit_bf16(p, A2, st, X[l], z[l], perms, batch_size=64)
            else:
                O.fit(p, A2, st, X[l], z[l], perms, batch_size=64, l2=l2_oracle)
            ref = pack(p)
            # (bfloat16: a flipped rounding of an activation moves an update by a bf16 ulp of the
            # gradient; stated tolerance of tests/test_gpu_parity.py's bf16 fit test)
            tol = (2e-3 + 2e-2 * np.abs(ref)) if bf else (2e-4 + 2e-3 * np.abs(ref))
            err = np.abs(th[l] - ref) / tol
            worst_theta = max(worst_theta, float(err.max()))
            pe = unpack(th[l].copy(), D2, U2)
            if bf:
                # E epochs of bfloat16 steps from the same state: a flipped rounding changes the sign of
                # a near-zero gradient and with it a whole Adam step (lr) of that weight -- single
                # weights drift apart, the fitted FUNCTION does not: nearly all weights within the
                # stated tolerance, the two classifiers' losses on the data within 2 %
                theta_within.append(float(np.mean(err <= 1.0)))
                a_dev = O.forward_bf16(pe, A2[:-1] + ["linear"], X[l])
                a_ref = O.forward_bf16(p, A2[:-1] + ["linear"], X[l])
                # -- the FUNCTION, row by row: the two classifiers' logits on the training set,
                # in bfloat16 steps at the logit's magnitude (2^-7 relative: the output is rounded to bfloat16)
                ulp = 2.0 ** (np.floor(np.log2(np.maximum(np.maximum(np.abs(a_dev), np.abs(a_ref)), 2.0 ** -6))) - 7)
                bf16_logit_ulps.append(float(np.max(np.abs(a_dev - a_ref) / ulp)))
                # -- the out-of-tolerance weights by cause: (gate) the weight feeds or leaves a hidden unit that is
                # ON for some training row under one net and OFF under the other -- a ReLU gate flipped by a
                # rounding --, (step) it differs by no more than a handful of Adam steps (a near-zero gradient
                # whose sign a rounding flips moves the weight by lr the other way: 2 lr per such step), (other)
                hs_d = O.forward_bf16(pe, A2[:-1] + ["linear"], X[l], return_all=True)
                hs_r = O.forward_bf16(p, A2[:-1] + ["linear"], X[l], return_all=True)
                flipped = [np.zeros(D2, bool)] + [np.any((hd > 0) != (hr > 0), axis=0) for hd, hr in zip(hs_d[1:], hs_r[1:])]
                off, fan = 0, D2
                cause = np.zeros(len(ref), np.int8)            # 1 gate, 2 step, 3 other (0: within tolerance)
                diff = np.abs(th[l] - ref)
                for li, u in enumerate(U2):
                    gate_w = (flipped[li][:, None] | flipped[li + 1][None, :]).ravel()
                    gate = np.concatenate([gate_w, flipped[li + 1]])
                    seg = slice(off, off + fan * u + u)
                    bad = err[seg] > 1.0
                    cause[seg] = np.where(bad, np.where(gate, 1, np.where(diff[seg] <= 8e-3, 2, 3)), 0)
                    off += fan * u + u
                    fan = u
                for k, nm in ((1, "gate"), (2, "step"), (3, "other")):
                    bf16_causes[nm] += int(np.sum(cause == k))
                bf16_causes["weights"] += len(ref)
                l_dev = float(O.bce_with_logits(a_dev, z[l].reshape(-1, 1)).mean())
                l_ref = float(O.bce_with_logits(a_ref, z[l].reshape(-1, 1)).mean())
                fit_loss_rel.append(abs(l_dev - l_ref) / l_ref)
                # What is asserted (round 5): the FUNCTION -- every training row's logit within 8 bfloat16 steps of
                # the oracle's (measured: <= 4), the losses within 2 % -- and that the weights outside the tolerance
                # are the ones a flipped ReLU gate or a flipped near-zero gradient explains (measured: 656 of 659).
                # The share of weights inside the tolerance (measured 0.975 - 1.0) is recorded, with a floor far
                # enough below it that a rounding flipped on another box does not turn the suite red.
                n_bad = int(np.sum(err > 1.0))
                n_other = int(np.sum(cause == 3))
                assert bf16_logit_ulps[-1] <= 8.0 and fit_loss_rel[-1] <= 0.02, (name, it, l, bf16_logit_ulps[-1], fit_loss_rel[-1])
                assert n_other <= max(3, 0.1 * n_bad), (name, it, l, n_bad, n_other)
                assert theta_within[-1] >= 0.93, (name, it, l, theta_within[-1])
                # (0.97 -- the floor before round 5, what every run so far has measured above -- stays the
                # recorded EXPECTATION: a run below it passes but says so, so that a drift is seen before it reaches 0.93)
                if theta_within[-1] < 0.97:
                    import warnings
                    warnings.warn(f"{name} iteration {it} loop {l}: share of bfloat16-fit weights within tolerance "
                                  f"{theta_within[-1]:.4f} < 0.97 (expected 0.975 - 1.0; hard floor 0.93)")
            else:
                assert err.max() <= 1.0, (name, it, l, float(err.max()))
            Xc = sampling.uniform_candidates(5, 1, NS, lo, hi, model_index0=100 + l, draw_index=it)[0]
            pred = (O.forward_bf16(pe, A2, Xc) if bf else O.predict(pe, A2, Xc)).squeeze(axis=-1)
            starts = Xc[np.argpartition(-pred, kth=R - 1, axis=None)[:R]]
            results = O.maxima(pe, A2, bounds, num_starts=R, num_samples=NS, X_init=Xc, compute=compute,
                               transform=transform)
            n_iter += 1
            s_o, s_d = {tuple(r) for r in starts}, {tuple(r) for r in x0d[l]}
            start_overlap.append(len(s_o & s_d) / R)
            n_same_starts += s_o == s_d
            best = None
            for res in results:
                if (res.success or res.status == 1) and (best is None or res.fun < best.fun):
                    best = res
            n_none_ora += best is None
            n_none_dev += bestd[l] < 0
            x_eng = xbd[l] if bestd[l] >= 0 else rs.uniform(lo, hi)
            x_next[l] = x_eng
            if best is not None and bestd[l] >= 0:
                n_pick_same += bool(np.allclose(x_eng, best.x, rtol=0, atol=1e-5)
                                    or abs(fund[l, bestd[l]] - best.fun) <= 2e-6)
                vg = O.value_and_input_grad_bf16 if bf else None
                v_dev = (float(vg(pe, A2, x_eng[None, :], transform, True)[0][0]) if bf else
                         float(O.value_and_input_grad(pe, A2, x_eng[None, :], transform)[0][0]))
                regret.append(v_dev - float(best.fun))
                regret64.append(oracle_value64(pe, x_eng) - oracle_value64(pe, np.asarray(best.x)))
            by_start = {tuple(s): k for k, s in enumerate(starts)}
            for r in range(R):              # restart by restart, where both sides have the start
                k = by_start.get(tuple(x0d[l, r]))
                if k is None:
                    continue
                res = results[k]
                acc_o, acc_d = bool(res.success or res.status == 1), infod[l, r, 2] in (0, 1)
                n_rest += 1
                n_acc_agree += acc_o == acc_d
                key = f"{int(res.status)}/{int(infod[l, r, 2])}"
                status_pairs[key] = status_pairs.get(key, 0) + 1
                dfun.append(float(fund[l, r]) - float(res.fun))
                if acc_o and acc_d:
                    n_both += 1
                    n_both_same += bool(np.allclose(xd[l, r], res.x, rtol=0, atol=1e-5))
        X = np.concatenate([X, x_next[:, None, :]], axis=1)
        y = np.concatenate([y, objective(x_next)[:, None]], axis=1)
    dfun, regret, regret64 = np.abs(dfun), np.asarray(regret), np.asarray(regret64)
    meas = dict(
        models=L, iterations=T, restarts=R, compute=compute, worst_theta_error_over_tolerance=worst_theta,
        same_starts=[int(n_same_starts), n_iter], start_overlap_mean=float(np.mean(start_overlap)),
        acceptance_agrees=[int(n_acc_agree), n_rest], accepted_within_1e5=[int(n_both_same), n_both],
        status_pairs_oracle_device=status_pairs,
        bf16_fit_theta_share_within_tolerance=theta_within or None, bf16_fit_loss_relative_difference=fit_loss_rel or None,
        bf16_fit_worst_logit_difference_in_bf16_ulps=bf16_logit_ulps or None,
        bf16_fit_out_of_tolerance_weights_by_cause=bf16_causes if bf else None,
        pick_same_or_equally_good=[int(n_pick_same), n_iter], none_device=int(n_none_dev), none_oracle=int(n_none_ora),
        abs_dfun_q50_q90_q99_max=[float(q) for q in np.quantile(dfun, [0.5, 0.9, 0.99, 1.0])] if len(dfun) else None,
        pick_regret_under_oracle_objective=[float(q) for q in np.sort(regret)],
        pick_regret_under_float64_objective=[float(q) for q in np.sort(regret64)])
    print(f"\n[end-to-end, {name}, teacher-forced]", meas)
    record_measurement(f"end_to_end_teacher_forced_{name}", meas)
    return meas


def test_config3_fit_and_argmax_chain_against_the_oracle_teacher_forced(gpu):
    """BASELINE config 3 (16-D, 64-64-64-1, float32) end to end: the launch chain
    fit_kernel<3> (wide rounds) -> sample + screen -> lbfgsb_kernel_w8<3> / queue -> pick against
    oracle.fit + oracle.maxima and the reference's acceptance rule (bore/mixins.py:57-89)."""
    m = _wide_chain_teacher_forced("cfg3", 16, [64, 64, 64, 1], ["relu", "relu", "relu", "sigmoid"], "float32",
                                   L=2, T=2, R=64, NS=1024, E=100, N0=96, seed=33)
    n_it = m["same_starts"][1]
    assert m["start_overlap_mean"] >= 0.9
    assert len(m["pick_regret_under_float64_objective"]) >= 0.75 * n_it
    # the stated float32 tolerance for suggested candidates: the device's pick, valued by the float64
    # network, is within 1e-3 of the oracle's own pick (better or worse)
    assert np.all(np.abs(m["pick_regret_under_float64_objective"]) <= 1e-3)


@pytest.mark.parametrize("D,l2", [(16, None), (6, None), (10, 1e-4)])
def test_plugin_default_network_chain_against_the_oracle_teacher_forced(gpu, D, l2):
    """The network the reference's only in-repo caller builds: BORE(num_layers=2, num_units=32,
    activation="elu", transform="sigmoid", num_starts=5), gamma = 1/3 (bore/plugins/hpbandster/base.py:23-33) ->
    DenseSequential's fall-through (bore/models.py:16-19) -> D -> 32-32-32-1, elu x3, a LINEAR output under
    from_logits BCE (base.py:145-157).  Static shape 5 at 16 inputs; on 6 / 10 inputs the fit runs zero-padded on
    the static kernel (fit_padded) and the acquisition kernels take the input dimension at run time; with
    l2_factor the fit is the generic flavour's and the acquisition static.  fit -> sample + screen -> 5 restarts ->
    pick, iteration by iteration against oracle.fit + oracle.maxima (scipy on the float32 oracle network)."""
    acts = ["elu", "elu", "elu", "linear"]
    m = _wide_chain_teacher_forced(f"plugin_default_D{D}" + ("_l2" if l2 else ""), D, [32, 32, 32, 1], acts, "float32",
                                   L=4, T=3, R=5, NS=1024, E=100, N0=40, seed=70 + D, transform="sigmoid", gamma=1.0 / 3.0,
                                   l2=l2)
    n_it = m["same_starts"][1]
    assert m["worst_theta_error_over_tolerance"] <= 1.0
    assert m["start_overlap_mean"] >= 0.9
    # the device's pick, valued by the float64 network under the sigmoid transform, is within 1e-3 of the
    # oracle's own pick (the stated float32 tolerance for suggested candidates)
    assert len(m["pick_regret_under_float64_objective"]) >= 0.75 * n_it
    assert np.all(np.abs(m["pick_regret_under_float64_objective"]) <= 1e-3)


def test_config5_bf16_fit_and_argmax_chain_against_the_bf16_oracle_teacher_forced(gpu):
    """BASELINE config 5 (32-D, 128-128-1, bfloat16) end to end:
    fit_bf16_mfma_kernel<4> -> screen (bf16 MFMA) -> lbfgsb_kernel<4, bf16> / queue -> pick against the
    mixed-precision oracle (fit_bf16, forward_bf16, value_and_input_grad_bf16 under scipy).  A
    bfloat16 prediction has 8 bits: both sides' objectives are step functions of x whose steps sit
    in slightly different places, so restart-by-restart agreement is weak by construction; what is
    asserted is the fit, the overlap of the screened starts and the quality of the pick."""
    m = _wide_chain_teacher_forced("cfg5_bf16", 32, [128, 128, 1], ["relu", "relu", "sigmoid"], "bfloat16",
                                   L=2, T=2, R=64, NS=1024, E=100, N0=96, seed=55)
    n_it = m["same_starts"][1]
    assert m["start_overlap_mean"] >= 0.7
    # valued by the float64 network the device's pick is within a bfloat16 step (2^-8) of the oracle's
    if len(m["pick_regret_under_float64_objective"]):
        assert np.all(np.abs(m["pick_regret_under_float64_objective"]) <= 2.0 ** -7)
    assert m["none_device"] + len(m["pick_regret_under_float64_objective"]) >= n_it - m["none_oracle"]
