#!/bin/bash
# same-box A/B of library variants on the per-level pyramid kernel (2048 640x480 pyramids): tools/pd_ab.sh variants/lib_a.so ...
export DSDTM_PY_DIAG=1   # the DSDTM_* switches below exist in the diagnostic library only (build.py --diag)
cd "$(dirname "$0")/.."
# the product library is put back when the script ends (the variants only ever replace it for the duration of a run)
LIB=dsdtm_amd/csrc/libdsdtm_amd.so
cp -p "$LIB" "$LIB.orig" && trap 'mv -f "$LIB.orig" "$LIB"' EXIT
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so; touch dsdtm_amd/csrc/libdsdtm_amd.so
  for i in 1 2; do echo -n "$so: "; DSDTM_PYR_FUSED=0 python tools/kernels.py 2>/dev/null | grep pyrDown | cut -c1-110; done
done
