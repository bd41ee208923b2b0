#!/bin/bash
# Builds library variants from -D switches HERE (hipcc cross-compiles), for a same-box A/B on the GPU:
#   tools/ab_defs.sh base: wake1:SA_WAKEUP=1,SA_SLEEP_SEQ=8 ...     -> variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
mkdir -p variants
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  args=""
  IFS=',' read -ra D <<< "$defs"
  for d in "${D[@]}"; do [ -n "$d" ] && args="$args --define $d"; done
  python dsdtm_amd/csrc/build.py $args > /dev/null
  cp dsdtm_amd/csrc/libdsdtm_amd.so variants/lib_$name.so
  echo "built variants/lib_$name.so ($defs)"
done
python dsdtm_amd/csrc/build.py > /dev/null      # leave the default build in place
