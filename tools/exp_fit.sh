#!/bin/bash
# time the wide fits with each experimental build given (bore_amd/csrc/libbore_hip_<tag>.so); GPU box
cd "$(dirname "$0")/.." || exit 1
for t in "$@"; do
  BORE_LIB_PATH=$PWD/bore_amd/csrc/libbore_hip_$t.so timeout -k 10 120 python3 tools/ab_fit.py $t 2>&1 | grep -v amdgpu.ids | grep "shape[34]" || exit 1
done
