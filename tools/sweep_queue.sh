#!/bin/bash
# queue workgroup-count sweep of the wide restart launches (GPU box)
mkdir -p gpurun_out/r4
out=gpurun_out/r4/cfg_queue_sweep.txt
: > $out
run() { timeout -k 10 200 python tools/cfg_restarts.py "$@" 2>&1 | grep -v amdgpu.ids; }
for q in 2048 4096 8192; do BORE_LBFGSB_QUEUE=$q run cfg2 256 3 | sed "s/^/Q=$q /" >> $out || exit 1; done
for q in 256 512 2048 4096; do BORE_LBFGSB_QUEUE=$q run cfg3 256 3 | sed "s/^/Q=$q /" >> $out || exit 1; done
for q in 256 512 2048 4096; do BORE_LBFGSB_QUEUE=$q run cfg5 256 3 | sed "s/^/Q=$q /" >> $out || exit 1; done
for c in cfg2 cfg3 cfg5; do run $c 64 3 >> $out; run $c 1 3 >> $out; done
cat $out
