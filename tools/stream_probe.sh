#!/bin/bash
# How many single-wave 2-ms kernels on separate streams does the device run at once, as a function of
# GPU_MAX_HW_QUEUES and the number of streams?  (the engine's creation probe, BORE_ASYNC_DEBUG prints it)
for q in 4 8 16 32; do for w in 6 12 24; do
  GPU_MAX_HW_QUEUES=$q W=$w BORE_ASYNC_DEBUG=1 timeout -k 5 60 python3 - <<'PY' 2>&1 | grep "bore_engine" | sed "s/^/GPU_MAX_HW_QUEUES=$q workers=$w: /"
import os
import numpy as np
from bore_amd.engine import NativeEngine
e = NativeEngine(np.arange(8), async_loops=True, objective="branin01", epochs=5, num_samples=64, resident_wait_us=0,
                 worker_streams=int(os.environ["W"]))
PY
done; done
