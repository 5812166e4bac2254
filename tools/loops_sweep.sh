#!/bin/bash
# T(1, L): BO-iterations/s of one GPU over the number of loops (GPU box).  usage: tools/loops_sweep.sh [objective] L...
obj=${1:-native}; shift
mkdir -p gpurun_out/r4
for L in "$@"; do
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-configs --repeats 3 --survey-steps 0 --objective $obj --loops $L > gpurun_out/r4/loops_${obj}_$L.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("gpurun_out/r4/loops_${obj}_$L.json").read().strip().splitlines()[-1])
x=d["kernels"][0]; p=d["phases"].get("per_loop_iteration_us",{})
print("loops $L ($obj objective): %.0f it/s  ms/step %.3f  device us/iteration: fit %.0f lbfgsb %.0f; host launch->result %.0f; %s launches; host finalize %.3f ms/step"%(d["value"],d["ms_per_step"],p.get("fit",0),p.get("lbfgsb",0),p.get("host_launch_to_result",0),x["launches"], d["phases"]["host_finalize_ms_per_step"]))
PY
done
