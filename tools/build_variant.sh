#!/bin/bash
# Build a diagnostic / experimental variant of the library next to the shipped one (never loaded
# unless BORE_LIB_PATH names it):  tools/build_variant.sh <tag> [extra hipcc flags...]
#   tools/build_variant.sh stamps -DBORE_STAMPS      -> bore_amd/csrc/libbore_hip_stamps.so
tag=$1; shift
cd "$(dirname "$0")/../bore_amd/csrc" || exit 1
exec /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -shared -fPIC -ffp-contract=off \
  "$@" bore_all.hip -o libbore_hip_${tag}.so
