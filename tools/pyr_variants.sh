#!/bin/bash
# same-box A/B of library variants on the pyramid kernels: tools/pyr_variants.sh variants/lib_a.so ...
cd "$(dirname "$0")/.."
# the product library is put back when the script ends (the variants only ever replace it for the duration of a run)
LIB=dsdtm_amd/csrc/libdsdtm_amd.so
cp -p "$LIB" "$LIB.orig" && trap 'mv -f "$LIB.orig" "$LIB"' EXIT
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so; touch dsdtm_amd/csrc/libdsdtm_amd.so
  echo "== $so"
  python tools/pyr_ab.py 2>&1 | grep -v amdgpu.ids
done
