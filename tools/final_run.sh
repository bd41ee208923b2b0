#!/bin/bash
# Everything behind profiles/r03_* that describes the round's final binary, in one GPU call: tests, the driver-style and
# the default bench lines, secondary timings, the pose-refinement soak, rocprofv3 passes.
# Usage on the GPU box: tools/final_run.sh <outdir-under-gpurun_out>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
O=gpurun_out/${1:-final}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -2 $O/tests.log
tools/profile.sh $(basename $O)/prof > $O/prof.log 2>&1
tail -3 $O/prof.log
cp $O/prof/r03_bench_pmc.json profiles/r03_bench_pmc.json      # the bench lines below quote this binary's counters
timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/r03_bench.json.log 2> $O/bench.err; echo "bench (driver-style) rc=$?"
timeout -k 10 600 python bench.py --no-secondary > $O/r03_bench_500.json.log 2>> $O/bench.err; echo "bench (500 steps) rc=$?"
( timeout -k 10 120 python tools/detect_bench.py; timeout -k 10 200 python tools/latency.py; timeout -k 10 200 python tools/track_step.py; timeout -k 10 200 python tools/stamps.py ) 2>/dev/null | grep -v amdgpu > $O/r03_secondary.txt
timeout -k 10 300 python tools/soak_pose_opt.py 700 2>/dev/null | tail -5 > $O/r03_pose_opt_soak.txt
echo done
