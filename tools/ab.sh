#!/bin/bash
# Same-box A/B of prebuilt library variants: tools/ab.sh variants/lib_a.so variants/lib_b.so ...
# (each is copied over dsdtm_amd/csrc/libdsdtm_amd.so in turn; the last one stays in place)
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for so in "$@"; do
  cp "$so" dsdtm_amd/csrc/libdsdtm_amd.so
  touch dsdtm_amd/csrc/libdsdtm_amd.so
  for rep in 1 2; do
    echo "== $so run $rep" | tee -a gpurun_out/ab.log
    python bench.py --steps 200 --warmup 20 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms_avg'])" | tee -a gpurun_out/ab.log
  done
  if [ -n "$AB_CHECK" ]; then python -m pytest tests/test_sparse_align_gpu.py -x -q -m gpu -k "golden or config2 or random" 2>&1 | tail -2 | tee -a gpurun_out/ab.log; fi
  if [ -n "$AB_STAMPS" ]; then python tools/stamps.py 2>/dev/null | tail -6 | tee -a gpurun_out/ab.log; fi
done
