#!/bin/bash
# Rebuilds ONE kernel source of libdanhip.so (bf16 build) with extra defines and relinks — for diagnosis builds:
#   tools/build_one.sh conv_wgrad_rows.hip -DDANHIP_WGRAD_DIAG
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$1; shift
EXTRA=""
case "$SRC" in *_exact.hip) EXTRA="-ffp-contract=off";; esac
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-gpu-rdc -Wall -Wno-unused-function $EXTRA "$@" -c "$ROOT/dan_amd/csrc/$SRC" -o "$ROOT/dan_amd/csrc/_obj/$SRC.o"
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/dan_amd/libdanhip.so" "$ROOT"/dan_amd/csrc/_obj/*.o
echo "relinked $ROOT/dan_amd/libdanhip.so"
