#!/bin/bash
# same-box A/B of library variants on the pyramid kernels: tools/pyr_variants.sh variants/lib_a.so ...
cd "$(dirname "$0")/.."
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so; touch dsdtm_amd/csrc/libdsdtm_amd.so
  echo "== $so"
  python tools/pyr_ab.py 2>&1 | grep -v amdgpu.ids
done
