#!/bin/bash
# rocprofv3 passes behind profiles/r06_*: kernel trace + stats, then one counter group per pass (never --pmc
# together with other trace domains). Usage on the GPU box: tools/profile.sh <outdir-under-gpurun_out>
# Each "case" is one command line; cases: main (bench defaults: 8 launch streams), solo (one stream), n1000, n2000, the
# secondary kernels (tools/kernels.py: pyrDown, Align2D), one tracked frame (tools/track_step.py: pyramid, single-pair
# alignment, warp prelude, Align2D, pose refinement, detector) and the batched pose refinement (tools/pose_opt_bench.py).
# Optional second argument: the cases to run (default: all) — a gpurun call is limited to 20 minutes, so the passes are
# split over two calls and their summaries merged by tools/merge_profiles.py.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-prof}"
CASES="${2:-main solo n1000 n2000 kernels track trackone poseopt secondary}"
has() { [[ " $CASES " == *" $1 "* ]]; }
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
COMMON="--no-cpu --no-secondary"
declare -A CASE
CASE[main]="$REPO/bench.py --steps 200 --warmup 20 $COMMON"
CASE[solo]="$REPO/bench.py --steps 200 --warmup 20 --streams 1 $COMMON"
CASE[n1000]="$REPO/bench.py --patches 1000 --steps 20 --warmup 3 --streams 1 $COMMON"
CASE[n2000]="$REPO/bench.py --width 1280 --height 960 --patches 2000 --pairs 256 --steps 20 --warmup 3 --streams 1 $COMMON"
CASE[kernels]="$REPO/tools/kernels.py"
CASE[track]="$REPO/tools/track_step.py"
# round 6: the same tracked frame through ONE dsdtm_track_frame call (pyramid, Run, reprojection, FindMatchDirect, replay, pose refinement)
CASE[trackone]="$REPO/tools/track_frame_bench.py"
CASE[poseopt]="$REPO/tools/pose_opt_bench.py 4096 200 nolatency"
# the driver's command WITH its secondary entries: the launch sizes the line's Align2D / FindMatchDirect / detector / single-pair
# entries use (their roofline.traffic reads this case)
CASE[secondary]="$REPO/bench.py --steps 5 --warmup 2 --preroll 0 --no-cpu"
for c in $CASES; do
  mkdir -p "$OUT/$c"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$c/trace" -- python3 ${CASE[$c]} > "$OUT/$c/trace.log" 2>&1 || echo "trace $c failed"
  echo "traced $c"
done
for c in $CASES; do
  [ "$c" = main ] && continue
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$c/pmc_fetch" -- python3 ${CASE[$c]} > "$OUT/$c/pmc_fetch.log" 2>&1 || echo "pmc fetch $c failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$c/pmc_write" -- python3 ${CASE[$c]} > "$OUT/$c/pmc_write.log" 2>&1 || echo "pmc write $c failed"
  echo "counted $c"
done
for c in solo n1000 n2000 poseopt; do
  has $c || continue
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/$c/pmc_sq" -- python3 ${CASE[$c]} > "$OUT/$c/pmc_sq.log" 2>&1 || echo "pmc sq $c failed"
done
# FP64 instruction mix of the alignment kernel (bench.py's fp64 block: flops per launch = 64 x (ADD + MUL + 2 FMA + TRANS))
if has solo; then timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d "$OUT/solo/pmc_fp64" -- python3 ${CASE[solo]} > "$OUT/solo/pmc_fp64.log" 2>&1 || echo "pmc fp64 failed"; fi
# keep the merge-back small: the per-dispatch CSVs of the long runs are summarised on the box
cd "$REPO"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
# the raw per-dispatch CSVs (tens of MB) stay on the box: the summaries above are what is kept (gpurun merges <= 64 MiB back)
for c in $CASES; do rm -rf "$OUT/$c/trace" "$OUT/$c"/pmc_*/ ; done
echo profdone
