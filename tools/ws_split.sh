#!/bin/bash
# workspace kernel: launch time by iteration cap and pair count (level starts vs iterations; occupancy of the launch)
cd "$(dirname "$0")/.."
for it in 1 2 4 10; do for pairs in 256 1024; do
python bench.py --patches 1000 --pairs $pairs --iters $it --steps 20 --warmup 3 --streams 1 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=1000 pairs $pairs iters $it: kernel %.4f ms' % d['roofline']['kernel_ms_avg'])"
done; done
