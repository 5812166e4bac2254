#!/bin/bash
# Round-6 measurements of one build on the GPU box (through gpurun, from the repo root):
#   bash tools/profile_r6.sh <tag>
# writes gpurun_out/<tag>_*; what should be judged is copied into profiles/r6/ afterwards.
set -o pipefail
tag=${1:-r6}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
HEAD="--steps 20 --warmup 5 --cpu-seconds 0 --no-configs --repeats 1 --survey-steps 0"   # the driver's headline run, once
echo "== bench as the driver runs it"; timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 --detail-dir $O > $O/${tag}_bench_driver_line.json 2> $O/${tag}_bench_driver.err || exit 1
cp $O/bench_detail_n1.json $O/${tag}_bench_driver.json
tail -n 1 $O/${tag}_bench_driver_line.json | cut -c1-400; echo; tail -n 1 $O/${tag}_bench_driver_line.json | wc -c
echo "== bench, SURVEY workload (T = 100)"; timeout -k 10 300 python3 bench.py --steps 100 --warmup 3 --cpu-seconds 0 --no-configs --survey-steps 0 --detail-dir $O > /dev/null 2>/dev/null || exit 1
cp $O/bench_detail_n1.json $O/${tag}_bench_T100.json
echo "== rocprofv3 kernel trace: headline"
rm -rf $O/prof_${tag}_kt
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_kt --output-format csv -- python3 bench.py $HEAD --detail-dir $O > /dev/null 2> $O/${tag}_rocprof_kt.err || exit 1
cp $O/bench_detail_n1.json $O/${tag}_bench_under_rocprof.json
python3 tools/prof_summary.py $O/prof_${tag}_kt $O/${tag}_headline > $O/${tag}_kt_summary.txt; cat $O/${tag}_kt_summary.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $pmc: headline"
  rm -rf $O/prof_${tag}_$pmc
  timeout -k 10 300 rocprofv3 --pmc $pmc -d $O/prof_${tag}_$pmc --output-format csv -- python3 bench.py $HEAD > /dev/null 2> $O/${tag}_rocprof_$pmc.err || exit 1
  python3 tools/prof_summary.py $O/prof_${tag}_$pmc $O/${tag}_headline_$pmc | tail -6
done
echo "== rocprofv3 --pmc SQ: headline"
rm -rf $O/prof_${tag}_sq
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/prof_${tag}_sq --output-format csv -- python3 bench.py $HEAD > /dev/null 2> $O/${tag}_rocprof_sq.err || exit 1
python3 tools/prof_summary.py $O/prof_${tag}_sq $O/${tag}_headline_sq | tail -6
echo "== rocprofv3 --pmc SQ (second set): headline"
rm -rf $O/prof_${tag}_sq2
timeout -k 10 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_INSTS_VMEM -d $O/prof_${tag}_sq2 --output-format csv -- python3 bench.py $HEAD > /dev/null 2> $O/${tag}_rocprof_sq2.err && python3 tools/prof_summary.py $O/prof_${tag}_sq2 $O/${tag}_headline_sq2 | tail -6 || echo "(second SQ set not collected: see ${tag}_rocprof_sq2.err)"
# the other BASELINE configs (2, 3, 5): bench.py's `configs` leg under the profiler
CFG="--steps 2 --warmup 1 --cpu-seconds 0 --repeats 1 --survey-steps 0"
echo "== rocprofv3 kernel trace: configs 2 / 3 / 5"
rm -rf $O/prof_${tag}_cfg_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_cfg_kt --output-format csv -- python3 bench.py $CFG --detail-dir $O > /dev/null 2> $O/${tag}_rocprof_cfg_kt.err || exit 1
cp $O/bench_detail_n1.json $O/${tag}_bench_cfg_under_rocprof.json
python3 tools/prof_summary.py $O/prof_${tag}_cfg_kt $O/${tag}_configs > $O/${tag}_cfg_kt_summary.txt; cat $O/${tag}_cfg_kt_summary.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $pmc: configs"
  rm -rf $O/prof_${tag}_cfg_$pmc
  timeout -k 10 400 rocprofv3 --pmc $pmc -d $O/prof_${tag}_cfg_$pmc --output-format csv -- python3 bench.py $CFG > /dev/null 2> $O/${tag}_rocprof_cfg_$pmc.err || exit 1
  python3 tools/prof_summary.py $O/prof_${tag}_cfg_$pmc $O/${tag}_configs_$pmc | tail -16
done
echo "== rocprofv3 --pmc SQ: configs"
rm -rf $O/prof_${tag}_cfg_sq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/prof_${tag}_cfg_sq --output-format csv -- python3 bench.py $CFG > /dev/null 2> $O/${tag}_rocprof_cfg_sq.err || exit 1
python3 tools/prof_summary.py $O/prof_${tag}_cfg_sq $O/${tag}_configs_sq | tail -16
echo "== wide configs: restart launches (queue), 256 / 64 / 1 loops"
for c in cfg2 cfg3 cfg5; do for L in 256 64 1; do timeout -k 10 200 python3 tools/cfg_restarts.py $c $L 5 2>&1 | grep -v amdgpu.ids; done; done | tee $O/${tag}_cfg_restarts.txt
python3 -c "import bench; print('csrc_sha256', bench.csrc_digest())" | tee $O/${tag}_csrc_digest.txt
echo "== loops per GPU"
for L in 1 64 128 256 512 768 1024 2048 4096 8192; do
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 3 --cpu-seconds 0 --no-configs --repeats 3 --survey-steps 0 --loops $L --detail-dir $O > /dev/null 2>/dev/null || exit 1
  cp $O/bench_detail_n1.json $O/${tag}_loops_$L.json
  python3 - <<PY
import json
d=json.loads(open("$O/${tag}_loops_$L.json").read().strip().splitlines()[-1])
x=d["kernels"][0]; p=d["phases"].get("per_loop_iteration_us",{})
print("loops $L: %.0f it/s  ms/step %.3f  device us/iteration: fit %.0f lbfgsb %.0f; host launch->result %.0f; %s launches"%(d["value"],d["ms_per_step"],p.get("fit",0),p.get("lbfgsb",0),p.get("host_launch_to_result",0),x["launches"]))
PY
done
rm -rf $O/prof_${tag}_*/*/*.db 2>/dev/null
du -sh $O | tail -1
