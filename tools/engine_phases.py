"""Where the restart phase of the fused iteration kernel spends its cycles, summed over EVERY
problem of a bench-shaped run (diagnostic build libbore_hip_stamps.so = -DBORE_STAMPS; GPU box):
per-routine cycles and calls (LDS accumulators, flushed per problem), the whole advance / f-g time,
per-problem totals (mean, p90, max) and the slowest problem of each loop.
usage: python tools/engine_phases.py [loops] [steps] [warmup] [formk]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_stamps.so"))
import numpy as np, torch
from bore_amd import _lib
from bore_amd.engine import NativeEngine
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = NativeEngine(np.arange(L), async_loops=True)
eng.run(warm)
eng.take_stats()
lib.bore_debug_lphases_reset()
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = eng.take_stats()
pp = (C.c_ulonglong * (4096 * 64))()
lib.bore_debug_lpp(pp)
pp = np.array(pp, dtype=np.float64).reshape(4096, 64)[:4 * L].reshape(L, 4, 64)[:, :3]   # 3 restarts per loop
names = ["cauchy", "formk", "cmprlb", "subsm", "lnsrlb", "matupd", "formt", "head", "freev", "accept", "cachechk", "bfgspair", "d=z-x"]
GAPS = {16: "two-variable search: evaluation -> dcsrch", 17: "two-variable search: dcsrch", 15: "outside advance (evaluation + kernel loop)", 18: "tail before return", 19: "gap before formt", 20: "gap entry->head", 21: "gap before cauchy", 22: "gap before freev",
        23: "gap before formk", 24: "gap before cmprlb", 25: "gap before subsm", 26: "gap before d=z-x", 27: "gap before lnsrlb",
        28: "gap before cachechk", 29: "gap before accept", 30: "gap before bfgspair", 31: "gap before matupd"}
if len(sys.argv) > 4 and sys.argv[4] == "formk":   # libbore_hip_stamps.so built with -DBORE_STAMPS_FORMK
    GAPS = {16: GAPS[16], 17: GAPS[17], 15: "outside advance + all gaps", 18: GAPS[18],
            19: "formk: new rows / column (after an update)", 20: "formk: old parts (entered / left variables)", 21: "formk: assembly of WN",
            22: "formk: Cholesky of block (1,1)", 23: "formk: diagonal check + triangular solves", 24: "formk: block (2,2)",
            25: "formk: Cholesky of block (2,2)", 26: "dcsrch: entry -> dcstep", 27: "dcsrch: dcstep"}
n_prob = L * 3 * steps
tot_adv, tot_fg = pp[..., 13].sum(), pp[..., 14].sum()
print(f"{L} loops x {steps} steps (stamps build): {L * steps / dt:.0f} it/s; device us per loop-iteration: "
      + " ".join(f"{k} {1e-3 * st['phase_ns_' + k] / max(st['phase_iterations'], 1):.1f}" for k in ("labels", "fit", "screen", "lbfgsb")))
print(f"per problem (mean over {n_prob}): advance {tot_adv / n_prob:.0f} cycles, f/g {tot_fg / n_prob:.0f}, sum {(tot_adv + tot_fg) / n_prob:.0f}")
acc = 0.0
for i, nm in enumerate(names):
    cyc, calls = pp[..., i].sum(), pp[..., 32 + i].sum()
    acc += cyc
    print(f"  {nm:7s}: {cyc / n_prob:9.0f} cycles per problem ({100 * cyc / (tot_adv + tot_fg):5.1f} %), {calls / n_prob:6.2f} calls, {cyc / max(calls, 1):7.0f} per call")
for g, nm in GAPS.items():
    cyc, calls = pp[..., g].sum(), pp[..., 32 + g].sum()
    print(f"  [{nm}]: {cyc / n_prob:9.0f} cycles per problem, {calls / n_prob:6.2f} times, {cyc / max(calls, 1):7.0f} each")
print(f"  rest of advance (saves, cache check, projgr, freev, tests, stamps): {(tot_adv - acc) / n_prob:9.0f} cycles per problem ({100 * (tot_adv - acc) / (tot_adv + tot_fg):5.1f} %)")
print(f"  rounds per problem: {pp[..., 32 + 13].sum() / n_prob:.1f}")
print(f"  f/g: {tot_fg / n_prob:9.0f} cycles per problem ({100 * tot_fg / (tot_adv + tot_fg):5.1f} %), {st['n_fg_rows'] / n_prob:.1f} evaluations, {tot_fg / max(st['n_fg_rows'], 1):.0f} per evaluation")
per_problem = (pp[..., 13] + pp[..., 14]) / steps           # mean cycles per iteration of this (loop, restart)
per_loop_max = per_problem.max(axis=1)
print(f"per (loop, restart) mean cycles per iteration: mean {per_problem.mean():.0f} p90 {np.quantile(per_problem, .9):.0f} max {per_problem.max():.0f}; "
      f"slowest restart of a loop: mean {per_loop_max.mean():.0f} p90 {np.quantile(per_loop_max, .9):.0f} max {per_loop_max.max():.0f}")
