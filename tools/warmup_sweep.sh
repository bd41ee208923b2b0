#!/bin/bash
# How much of the gap between a 20-step run and the 500-step steady state is the GPU coming out of idle:
# the same 20 timed steps behind more and more warm-up launches, then longer timed regions.
set -u
OUT=${1:-gpurun_out/warmup_sweep}
mkdir -p "$OUT"
for cfg in "20 5" "20 50" "20 200" "20 1000" "100 5" "500 5" "500 50" "20 5"; do
  set -- $cfg
  timeout -k 10 120 python bench.py --steps $1 --warmup $2 --no-cpu --no-secondary --preroll ${PREROLL:-0} 2>/dev/null |
    python -c "import json,sys; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('steps %4d warmup %4d: %.3f M alignments/s, span %.4f ms/step, solo %.4f ms' % (d['steps'], d['warmup'], d['value']/1e6, r['span_ms_per_step'], r['kernel_ms_solo']))" | tee -a "$OUT/sweep.txt" || exit 1
done
