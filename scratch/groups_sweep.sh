#!/bin/bash
# throughput vs number of stream groups / hardware queues (native engine)
for q in 8 12 16 24 32; do for g in 3 4 5 6 8 12; do
  [ $g -ge $q ] && continue
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python bench.py --steps 60 --warmup 3 --cpu-seconds 0 --groups $g > gpurun_out/sweep_${g}_$q.json 2>gpurun_out/sweep_${g}_$q.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_${g}_$q.json").read().strip().splitlines()[-1])
k={x["kernel"]:x["avg_launch_ms"] for x in d["kernels"]}
print("queues $q groups $g: %.0f it/s, ms/step %.3f, fit %.3f lbfgsb %.3f, host enq %.3f fin %.3f"%(d["value"],d["ms_per_step"],k["fit_kernel"],k["lbfgsb_kernel"],d["phases"]["host_enqueue_ms_per_step"],d["phases"]["host_finalize_ms_per_step"]))
PY
done; done
