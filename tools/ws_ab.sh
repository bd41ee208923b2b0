#!/bin/bash
# workspace kernel: one workgroup of eight waves per CU against two of four (DSDTM_WS_WAVES), windows on/off
cd "$(dirname "$0")/.."
run() { python bench.py "$@" --steps 20 --warmup 3 --streams 1 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: kernel %.4f ms  %.0f /s  parity %s' % ('$LABEL', d['roofline']['kernel_ms_avg'], d['value'], d.get('pose_delta_vs_cpu')))"; }
for w in 8 4; do
  export DSDTM_WS_WAVES=$w
  LABEL="waves $w N=1000 x1024"; run --patches 1000
  LABEL="waves $w N=720 x1024"; run --patches 720
  LABEL="waves $w N=2000 1280x960 x256"; run --width 1280 --height 960 --patches 2000 --pairs 256
  LABEL="waves $w N=1000 x1024 streams4"; python bench.py --patches 1000 --steps 40 --warmup 4 --streams 4 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL: %.0f /s' % d['value'])"
done
