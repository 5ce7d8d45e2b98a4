#!/bin/bash
# Same-box A/B of kernel changes: builds dan_amd/libdanhip_<name>.so = the CURRENT objects with the named csrc files compiled as they were at
# <git-ref> (e.g. HEAD~1) - select it with DANHIP_LIB_PATH=dan_amd/libdanhip_<name>.so (dan_amd/_lib.py); tools/ab_bench.py alternates the two.
#   tools/ab_variant.sh prev HEAD~1 loss.hip [conv_halo.hip ...]
#   AB_DEFINES=-DH_ABLATE_EPILOGUE tools/ab_variant.sh noepi WORK conv_halo.hip      (WORK = the working tree's file, with extra defines: ablation builds)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; REF=$2; shift 2
python3 -m dan_amd.build > /dev/null
TMP=$(mktemp -d)
mkdir -p "$TMP/csrc" "$TMP/include" "$TMP/obj"
cp "$ROOT"/dan_amd/csrc/*.h "$TMP/csrc/"; mkdir -p "$TMP/../include" 2>/dev/null || true
cp "$ROOT"/dan_amd/csrc/_obj/*.o "$TMP/obj/"
# headers as they were at the ref too (a source may depend on its header version)
mkdir -p "$TMP/inc2/include"
if [ "$REF" = "WORK" ]; then cp "$ROOT/include/danhip.h" "$TMP/inc2/include/danhip.h"; else
  for h in $(git -C "$ROOT" ls-tree --name-only "$REF" dan_amd/csrc/ | grep '\.h$'); do git -C "$ROOT" show "$REF:$h" > "$TMP/csrc/$(basename $h)"; done
  git -C "$ROOT" show "$REF:include/danhip.h" > "$TMP/inc2/include/danhip.h"
fi
for SRC in "$@"; do
  if [ "$REF" = "WORK" ]; then cp "$ROOT/dan_amd/csrc/$SRC" "$TMP/csrc/$SRC"; else git -C "$ROOT" show "$REF:dan_amd/csrc/$SRC" > "$TMP/csrc/$SRC"; fi
  # the sources include "../../include/danhip.h": give them that relative layout
  mkdir -p "$TMP/t/dan_amd/csrc" "$TMP/t/include"
  cp "$TMP"/csrc/* "$TMP/t/dan_amd/csrc/"; cp "$TMP/inc2/include/danhip.h" "$TMP/t/include/"
  EXTRA=""; case "$SRC" in *_exact.hip) EXTRA="-ffp-contract=off";; esac
  XH=""; case "$SRC" in *.cpp) XH="-x hip";; esac
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -munsafe-fp-atomics -fno-gpu-rdc -Wno-unused-function $EXTRA $AB_DEFINES $XH -c "$TMP/t/dan_amd/csrc/$SRC" -o "$TMP/obj/$SRC.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/dan_amd/libdanhip_$NAME.so" "$TMP"/obj/*.o
rm -rf "$TMP"
echo "built $ROOT/dan_amd/libdanhip_$NAME.so ($* from $REF)"
