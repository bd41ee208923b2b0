#!/bin/bash
# tools/ab_sizes.sh lib_a.so lib_b.so ... : bench at several batch sizes for each variant (same box)
set -e
cd "$(dirname "$0")/.."
# the product library is put back when the script ends (the variants only ever replace it for the duration of a run)
LIB=dsdtm_amd/csrc/libdsdtm_amd.so
cp -p "$LIB" "$LIB.orig" && trap 'mv -f "$LIB.orig" "$LIB"' EXIT
mkdir -p gpurun_out
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so; touch dsdtm_amd/csrc/libdsdtm_amd.so
  for n in 512 1024 2048 4096; do
    echo -n "$so pairs=$n: " | tee -a gpurun_out/ab_sizes.log
    python bench.py --steps 100 --warmup 10 --no-cpu --pairs $n | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), round(d['roofline']['kernel_ms_avg'],4))" | tee -a gpurun_out/ab_sizes.log
  done
done
