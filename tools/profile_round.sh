#!/bin/bash
# Measurements of one build on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh v10
# writes gpurun_out/<tag>_*; copy what should be judged into profiles/r1/.
set -o pipefail
tag=${1:-vX}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
export TMPDIR=/tmp
cd $R
echo "== bench (default run)"; timeout -k 10 400 python3 bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || exit 1
tail -c 600 $O/${tag}_bench.json; echo
echo "== rocprofv3 kernel trace"
rm -rf $O/prof_${tag}_kt
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $O/prof_${tag}_kt --output-format csv -- python3 bench.py --steps 100 --warmup 3 --cpu-seconds 0 > $O/${tag}_bench_under_rocprof.json 2> $O/${tag}_rocprof_kt.err || exit 1
python3 tools/prof_summary.py $O/prof_${tag}_kt $O/${tag} > $O/${tag}_kt_summary.txt; cat $O/${tag}_kt_summary.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  echo "== rocprofv3 --pmc $pmc"
  rm -rf $O/prof_${tag}_$pmc
  timeout -k 10 400 rocprofv3 --pmc $pmc -d $O/prof_${tag}_$pmc --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --loops 512 > /dev/null 2> $O/${tag}_rocprof_$pmc.err || exit 1
  python3 tools/prof_summary.py $O/prof_${tag}_$pmc $O/${tag}_$pmc | tail -12
done
echo "== rocprofv3 --pmc SQ"
rm -rf $O/prof_${tag}_sq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/prof_${tag}_sq --output-format csv -- python3 bench.py --steps 4 --warmup 1 --cpu-seconds 0 --loops 512 > /dev/null 2> $O/${tag}_rocprof_sq.err || exit 1
python3 tools/prof_summary.py $O/prof_${tag}_sq $O/${tag}_sq | tail -12
echo "== lock-step groups schedule"; timeout -k 10 300 python3 bench.py --cpu-seconds 0 --schedule groups > $O/${tag}_bench_groups.json 2>/dev/null; tail -c 300 $O/${tag}_bench_groups.json | head -c 10; python3 -c "import json;d=json.loads(open('$O/${tag}_bench_groups.json').read().strip().splitlines()[-1]);print('groups schedule: %.0f it/s'%d['value'])"
echo "== configs 2/3/5, one model"; timeout -k 10 300 python3 tools/cfg_time.py > $O/${tag}_cfg_time.txt 2>&1; cat $O/${tag}_cfg_time.txt
echo "== loops per GPU"
for L in 64 128 256 512 1024 2048 4096; do
  timeout -k 10 200 python3 bench.py --steps 40 --warmup 3 --cpu-seconds 0 --loops $L > $O/${tag}_loops_$L.json 2>/dev/null || exit 1
  python3 - <<PY
import json
d=json.loads(open("$O/${tag}_loops_$L.json").read().strip().splitlines()[-1])
k={x["kernel"]:x for x in d["kernels"]}
x=d["kernels"][0]
print("loops $L: %.0f it/s  ms/step %.3f  %s %.3f ms x %d launches (%.1f in flight, %.0f GB/s algorithmic in aggregate)"%(d["value"],d["ms_per_step"],x["kernel"],x["avg_launch_ms"],x["launches"],x["share_of_step"],x["aggregate_GBs"]))
PY
done
rm -rf $O/prof_${tag}_*/*/*.db 2>/dev/null
du -sh $O | tail -1
