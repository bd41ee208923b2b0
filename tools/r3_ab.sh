#!/bin/bash
# Same-box A/B of the <= 320-feature kernel with two and with three pair slots per compute unit (Options::reg_slots,
# read from DSDTM_REG_SLOTS at dsdtm_create). Usage on the GPU box: tools/r3_ab.sh <outdir-under-gpurun_out> [reps]
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
O=gpurun_out/${1:-r3ab}; mkdir -p $O
for rep in $(seq 1 ${2:-2}); do
  for slots in 2 3; do
    for st in 1 8; do
      DSDTM_REG_SLOTS=$slots timeout -k 10 200 python bench.py --steps 300 --warmup 30 --no-cpu --no-secondary --streams $st | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $slots streams $st: %.0f /s  kernel %.4f ms  ms/step %.4f' % (d['value'], d['roofline']['kernel_ms_avg'], d['ms_per_step']))" | tee -a $O/ab.txt
    done
  done
done
