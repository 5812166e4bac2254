#!/bin/bash
# Build libbore_hip.so from any directory (bore_amd._lib.build_native: hipcc for gfx950, the sources' digest compiled
# in); with "dbg" also the -DBORE_WIDE_STAMPS diagnostic build.
R="$(cd "$(dirname "$0")/.." && pwd)"
cd "$R" && python3 -c "from bore_amd import _lib; print(_lib.build_native(force=True, verbose=False))" 2>/tmp/build_main.err || { grep -A6 " error" /tmp/build_main.err | head -40; exit 1; }
if [ "$1" = "dbg" ]; then
  F="-O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -ffp-contract=off"
  /opt/rocm/bin/hipcc $F -DBORE_WIDE_STAMPS "$R/bore_amd/csrc/bore_all.hip" -o "$R/bore_amd/csrc/libbore_hip_dbg.so" 2>/tmp/build_dbg.err || exit 1
fi
ls -la "$R"/bore_amd/csrc/*.so
