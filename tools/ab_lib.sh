#!/bin/bash
# Same-box A/B of prebuilt library variants with any tool: tools/ab_lib.sh "python tools/ws_cap.py" variants/lib_a.so variants/lib_b.so ...
# Each variant is copied over the product library for the duration of its run (label = the variant's file name); the
# product library is restored at the end.
cd "$(dirname "$0")/.."
CMD="$1"; shift
LIB=dsdtm_amd/csrc/libdsdtm_amd.so
cp -p "$LIB" "$LIB.orig" && trap 'mv -f "$LIB.orig" "$LIB"' EXIT
for so in "$@"; do
  cp "$so" "$LIB" && touch "$LIB"
  timeout -k 10 300 $CMD "$(basename "$so" .so)" || exit 1
done
