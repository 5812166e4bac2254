#!/bin/bash
# throughput and per-kernel times vs number of stream groups / hardware queues
for cfg in "4 8" "8 8" "8 16" "16 16" "16 24" "32 32"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=$2 timeout -k 10 120 python bench.py --steps 60 --warmup 3 --cpu-seconds 0 --groups $1 > gpurun_out/sweep_$1_$2.json 2>gpurun_out/sweep_$1_$2.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/sweep_$1_$2.json").read().strip().splitlines()[-1])
k={x["kernel"]:x["avg_launch_ms"] for x in d["kernels"]}
print("groups $1 queues $2: %.0f it/s, ms/step %.3f, fit %.3f lbfgsb %.3f, host enq %.3f fin %.3f"%(d["value"],d["ms_per_step"],k["fit_kernel"],k["lbfgsb_kernel"],d["phases"]["host_enqueue_ms_per_step"],d["phases"]["host_finalize_ms_per_step"]))
PY
done
