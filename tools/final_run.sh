#!/bin/bash
# Everything behind profiles/r02_* in one GPU call: tests, the default bench line, rocprofv3 passes, stamps,
# latencies. Usage on the GPU box: tools/final_run.sh <outdir-under-gpurun_out>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
cd "$REPO"
O=gpurun_out/${1:-final}; mkdir -p $O
python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -2 $O/tests.log
python bench.py > $O/r02_bench.json.log 2> $O/bench.err; echo "bench rc=$?"
python bench.py --streams 1 --no-secondary > $O/r02_bench_one_stream.json.log 2>> $O/bench.err
tools/streams_sweep.sh $(basename $O)/sweep > $O/r02_streams_sweep.txt 2>&1
python tools/stamps.py > $O/r02_stamps.txt 2>/dev/null
python tools/latency_by_count.py > $O/r02_latency_by_count.txt 2>/dev/null
( python tools/kernels.py; python tools/latency.py; python tools/cpp_detect.py ) 2>/dev/null | grep -v amdgpu > $O/r02_secondary_kernels.txt
python tools/track_step.py > $O/r02_track_step.txt 2>/dev/null
( PYR_SMALL=1 PYR_ALL=1 python tools/pyr_ab.py ) 2>/dev/null | grep -v amdgpu > $O/r02_pyramid_ab.txt
( python tools/pose_opt_bench.py 1 200 nolatency; python tools/pose_opt_bench.py 64 200; python tools/pose_opt_bench.py 4096 200 nolatency ) 2>/dev/null | grep -v amdgpu > $O/r02_pose_opt.txt
( tools/ws_ab.sh; tools/ws_split.sh ) 2>/dev/null | grep -v amdgpu > $O/r02_workspace_kernel.txt
python tools/soak_pose_opt.py 700 2>/dev/null | tail -4 > $O/r02_pose_opt_soak.txt
tools/profile.sh $(basename $O)/prof > $O/prof.log 2>&1
tail -3 $O/prof.log
echo done
