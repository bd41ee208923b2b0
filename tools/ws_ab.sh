#!/bin/bash
# The shapes that run on the workspace kernel (LDS-parked grid inputs; one pair per compute unit up to 1024 patches, one
# pair on two compute units up to 2048), one launch at a time and on 4 streams, plus the headline shape as a control:
# one line per shape, for same-box comparisons of two builds (run it once per build in the same gpurun call).
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
O=gpurun_out/${1:-wsab}; mkdir -p $O
run() { timeout -k 10 120 python bench.py "$@" --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: %.4f ms per step  %.0f /s' % ('$LABEL', d['ms_per_step'], d['value']))" || exit 1; }
for rep in 1 2; do
  LABEL="${2:-build} N=1000 640x480 x1024 one stream";  run --patches 1000 --steps 30 --warmup 10 --streams 1 | tee -a $O/ab.txt
  LABEL="${2:-build} N=1000 640x480 x1024 4 streams";   run --patches 1000 --steps 60 --warmup 10 --streams 4 | tee -a $O/ab.txt
  LABEL="${2:-build} N=2000 1280x960 x256 one stream";  run --width 1280 --height 960 --patches 2000 --pairs 256 --steps 30 --warmup 10 --streams 1 | tee -a $O/ab.txt
  LABEL="${2:-build} N=2000 1280x960 x256 4 streams";   run --width 1280 --height 960 --patches 2000 --pairs 256 --steps 60 --warmup 10 --streams 4 | tee -a $O/ab.txt
  LABEL="${2:-build} N=720 640x480 x1024 one stream";   run --patches 720 --steps 30 --warmup 10 --streams 1 | tee -a $O/ab.txt
  LABEL="${2:-build} N=300 640x480 x1024 8 streams";    run --steps 200 --warmup 20 | tee -a $O/ab.txt
done
