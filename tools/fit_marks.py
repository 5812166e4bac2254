"""Where an Adam step of the 2->16-16-1 fit spends its cycles inside the fused iteration kernel, summed
over every step of every loop of a bench-shaped run (diagnostic build libbore_hip_fitmarks.so =
-DBORE_FIT_MARKS; GPU box).  usage: python tools/fit_marks.py [loops] [steps] [warmup]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BORE_LIB_PATH", os.path.join(ROOT, "bore_amd", "csrc", "libbore_hip_fitmarks.so"))
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
buf = (C.c_ulonglong * 256)()
lib.bore_debug_fit_marks(buf, 1)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.run(steps)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = eng.take_stats()
lib.bore_debug_fit_marks(buf, 0)
a = np.array(buf, dtype=np.float64).reshape(8, 32)
names = ["gather+requests", "forward", "loss+delta", "backward+copies", "wait mid barrier", "dW+Adam phase", "wait end barrier",
         "step loop top -> step", "  task: requests", "  task: matrix chain", "  task: Adam+stores", "(a mark itself)", "(end barrier -> epoch top)", "(epoch top -> shuffle chosen)", "(-> step loop top)", "(mid barrier -> own task done)"]
n_steps = a[:, 16 + 5].max()          # every wave passes mark 5 once per Adam step
print(f"{L} loops x {steps} steps (marks build): {L * steps / dt:.0f} it/s; fit {1e-3 * st['phase_ns_fit'] / max(st['phase_iterations'], 1):.1f} us per loop-iteration; "
      f"{n_steps:.0f} Adam steps marked, N {eng.N - steps}..{eng.N - 1}")
print("cycles per Adam step, per wave (mean over the steps in which the wave passed the mark; share of steps):")
for i, nm in enumerate(names):
    row = []
    for w in range(4):
        c, k = a[w, i], a[w, 16 + i]
        row.append(f"{c / max(k, 1):7.0f} ({k / max(n_steps, 1):4.2f})")
    print(f"  {nm:22s} " + "  ".join(row))
tot = [sum(a[w, i] for i in range(8)) / max(n_steps, 1) for w in range(4)]
print("  sum of 0..7 per step   " + "  ".join(f"{t:7.0f}       " for t in tot))
