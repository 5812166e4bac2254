#!/bin/bash
# T = 100 (SURVEY form) on experiment builds: throughput, launches, per-phase device time (GPU box)
mkdir -p gpurun_out/r4
for t in "$@"; do
  BORE_LIB_PATH=$PWD/bore_amd/csrc/libbore_hip_$t.so timeout -k 10 200 python bench.py --steps 100 --warmup 3 --repeats 3 --no-configs --cpu-seconds 0 --survey-steps 0 > gpurun_out/r4/survey_$t.json 2> gpurun_out/r4/survey_$t.err || { tail -3 gpurun_out/r4/survey_$t.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4/survey_$t.json").read().strip().splitlines()[-1])
k=d["kernels"][0]; p=d["phases"].get("per_loop_iteration_us",{})
print("$t: %.0f it/s, launches %d, avg launch %.2f ms, fit %.0f lbfgsb %.0f host l->r %.0f, loops/launch %.1f" % (d["value"], k["launches"], k["avg_launch_ms"], p.get("fit",0), p.get("lbfgsb",0), p.get("host_launch_to_result",0), d["phases"].get("loops_per_launch",0)))
PY
done
