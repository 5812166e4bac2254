"""Soak: the work-queue schedule of the asynchronous engine (more loops than resident workgroups; any
workgroup takes any loop's next iteration, so a loop's state moves between compute units and XCDs every
iteration) against the lock-step engine over 30 BO iterations of 1100 loops (SOAK_LOOPS, SOAK_STEPS to change); trajectories and weights
must be bit-equal.  GPU box: python tools/soak_queue.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from bore_amd.engine import NativeEngine
L = int(os.environ.get("SOAK_LOOPS", 1100))
seeds = np.arange(5000, 5000 + L)
a = NativeEngine(seeds, async_loops=True, objective="branin01")
b = NativeEngine(seeds, groups=3, objective="branin01")
t0 = time.time()
STEPS = tuple(int(v) for v in os.environ.get("SOAK_STEPS", "7,11,12").split(","))
for n in STEPS:
    a.run(n)
ta = time.time() - t0
b.run(sum(STEPS))
Xa, ya = a.observations(); Xb, yb = b.observations()
st = a.take_stats(reset=False)
print("loops", L, "launches", st["fit_launches"], "shape", Xa.shape, "equal X", np.array_equal(Xa, Xb),
      "equal y", np.array_equal(ya, yb),
      "state", all(np.array_equal(u, v) for u, v in zip(a.state(), b.state())),
      "async %.1f s, total %.1f s" % (ta, time.time() - t0))
a.close(); b.close()
