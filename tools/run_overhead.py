"""Fixed cost of one engine.run() call against its per-step cost (GPU box).
usage: python tools/run_overhead.py LOOPS"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from bore_amd.engine import NativeEngine
L = int(sys.argv[1])
eng = NativeEngine(np.arange(L), async_loops=True, objective="branin01")
eng.run(3)
torch.cuda.synchronize()
for steps in (1, 2, 5, 10, 20, 1, 20):
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{L} loops, run({steps}): {1e3 * dt:.2f} ms = {1e3 * dt / steps:.3f} ms per step")
