#!/bin/bash
# Same-box A/B of prebuilt library variants: tools/ab.sh variants/lib_a.so variants/lib_b.so ...
# (each is copied over dsdtm_amd/csrc/libdsdtm_amd.so in turn; the product library is restored at the end). Per variant: two runs
# with one launch stream (kernel time by HIP events) and two with the bench's default streams (throughput).
set -e
cd "$(dirname "$0")/.."
# the product library is put back when the script ends (the variants only ever replace it for the duration of a run)
LIB=dsdtm_amd/csrc/libdsdtm_amd.so
cp -p "$LIB" "$LIB.orig" && trap 'mv -f "$LIB.orig" "$LIB"' EXIT
mkdir -p gpurun_out
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so
  touch dsdtm_amd/csrc/libdsdtm_amd.so
  for st in 1 1 4 4; do
    python bench.py --steps 300 --warmup 30 --no-cpu --no-secondary --streams $st | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$so streams $st: %.0f /s  kernel %.4f ms' % (d['value'], d['roofline']['kernel_ms_avg']))" | tee -a gpurun_out/ab.log
  done
  if [ -n "$AB_CHECK" ]; then python -m pytest tests/test_sparse_align_gpu.py -x -q -m gpu -k "golden or config2 or random or deterministic" 2>&1 | tail -2 | tee -a gpurun_out/ab.log; fi
  if [ -n "$AB_STAMPS" ]; then python tools/stamps.py 2>/dev/null | tail -16 | tee -a gpurun_out/ab.log; fi
done
