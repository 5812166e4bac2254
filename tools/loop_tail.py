"""Why the K-step region of the headline engine lasts longer than the mean loop's device time: per-loop
sums over the timed steps (diagnostic export bore_debug_engine_loop_stats; GPU box).
usage: python tools/loop_tail.py [loops] [steps] [warmup]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from bore_amd import _lib
from bore_amd.engine import NativeEngine
lib = _lib.lib()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
warm = int(sys.argv[3]) if len(sys.argv) > 3 else 5
lib.bore_debug_engine_loop_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
eng = NativeEngine(np.arange(L), async_loops=True, objective="branin01")
eng.run(warm)
eng.take_stats()
lib.bore_debug_engine_loop_stats(eng._h, None, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
out = np.zeros((L, 6))
lib.bore_debug_engine_loop_stats(eng._h, out.ctypes.data_as(C.c_void_p), 0)
fit, rst, nfev, rounds, flight, its = out.T
tot = flight * 1e3          # ms per loop over the region (launch -> result, summed over its iterations)
print(f"{L} loops x {steps} steps: {L * steps / dt:.0f} it/s, region {dt * 1e3:.2f} ms; per loop (ms over the region): "
      f"launch->result mean {tot.mean():.2f} p50 {np.median(tot):.2f} p90 {np.percentile(tot, 90):.2f} p99 {np.percentile(tot, 99):.2f} max {tot.max():.2f}")
for nm, v in (("fit ms", fit * 1e-6), ("restarts ms", rst * 1e-6), ("evaluations", nfev), ("rounds of the slowest restart", rounds)):
    print(f"  {nm:30s} mean {v.mean():9.2f} p50 {np.median(v):9.2f} p90 {np.percentile(v, 90):9.2f} p99 {np.percentile(v, 99):9.2f} max {v.max():9.2f}")
c = np.corrcoef(np.vstack([tot, fit, rst, nfev, rounds]))
print("  correlation of a loop's launch->result time with: fit %.2f, restarts %.2f, evaluations %.2f, slowest-restart rounds %.2f" % tuple(c[0, 1:]))
print("  restart us per round of the slowest restart, by decile of the loop's time:",
      " ".join("%.2f" % (1e-3 * rst[idx].sum() / max(rounds[idx].sum(), 1)) for idx in np.array_split(np.argsort(tot), 10)))
print("  fit us per iteration, by decile of the loop's time:",
      " ".join("%.0f" % (1e-3 * fit[idx].sum() / its[idx].sum()) for idx in np.array_split(np.argsort(tot), 10)))
slow = np.argsort(tot)[-8:]
print("  slowest loops:", " ".join(f"{l}:{tot[l]:.1f}ms(fit {fit[l]*1e-6:.1f}, rst {rst[l]*1e-6:.1f}, rounds {rounds[l]:.0f})" for l in slow))
