#!/bin/bash
# Where the time goes, two against three pair slots: launch time by iteration cap (cap 1 = level starts + one pass per level)
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
O=gpurun_out/${1:-r3split}; mkdir -p $O
for slots in 2 3; do
  for it in 1 2 4 10; do
    DSDTM_REG_SLOTS=$slots timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu --no-secondary --streams 1 --iters $it | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slots $slots cap $it: kernel %.4f ms  executed iterations per pair %.2f' % (d['roofline']['kernel_ms_avg'], d['executed_iterations_total_mean']))" | tee -a $O/split.txt
  done
done
