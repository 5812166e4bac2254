#!/bin/bash
# Build libbore_hip.so (and, with "dbg", the -DBORE_WIDE_STAMPS diagnostic build) from any directory.
R="$(cd "$(dirname "$0")/.." && pwd)"
F="-O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -ffp-contract=off"
/opt/rocm/bin/hipcc $F "$R/bore_amd/csrc/bore_all.hip" -o "$R/bore_amd/csrc/libbore_hip.so" 2>/tmp/build_main.err || { grep -A6 " error" /tmp/build_main.err | head -40; exit 1; }
if [ "$1" = "dbg" ]; then
  /opt/rocm/bin/hipcc $F -DBORE_WIDE_STAMPS "$R/bore_amd/csrc/bore_all.hip" -o "$R/bore_amd/csrc/libbore_hip_dbg.so" 2>/tmp/build_dbg.err || exit 1
fi
ls -la "$R"/bore_amd/csrc/*.so
