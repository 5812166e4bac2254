"""Where a loop-iteration's time goes under the asynchronous schedule, for several loop counts:
device-clock phases of the fused kernel vs the host's view (GPU box).
usage: python tools/async_diag.py [steps] [L ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bore_amd.engine import NativeEngine

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
Ls = [int(a) for a in sys.argv[2:]] or [1, 8, 64, 256, 512, 1024]
for L in Ls:
    eng = NativeEngine(np.arange(L), async_loops=True)
    eng.run(3)
    eng.take_stats()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.run(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    st = eng.take_stats()
    n = max(st["phase_iterations"], 1)
    dev = {k: 1e-3 * st["phase_ns_" + k] / n for k in ("labels", "fit", "screen", "lbfgsb")}
    print(f"L={L:5d}: {L * steps / dt:9.0f} it/s, {1e3 * dt / steps:6.3f} ms/step | device us: "
          + " ".join(f"{k} {v:6.1f}" for k, v in dev.items()) + f" sum {sum(dev.values()):7.1f} | host us: "
          f"ready->launch {1e6 * st['ready_to_launch_s'] / n:6.1f} launch->result {1e6 * st['launch_to_result_s'] / n:7.1f} "
          f"result->ready {1e6 * st['result_to_ready_s'] / n:6.1f} | {n / max(st['batches'], 1):6.1f} loops/launch, "
          f"kernel {st['argmax_ms'] / max(st['argmax_launches'], 1):.3f} ms x {st['argmax_launches']}, "
          f"fg rows/it {st['n_fg_rows'] / n:.0f}", flush=True)
    del eng
