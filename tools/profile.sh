#!/bin/bash
# rocprofv3 passes behind profiles/r03_*: kernel trace + stats, then one counter group per pass (never --pmc
# together with other trace domains). Usage on the GPU box: tools/profile.sh <outdir-under-gpurun_out>
# Each "case" is one command line; cases: main (bench defaults: 8 launch streams), solo (one stream), n1000, n2000, the
# secondary kernels (tools/kernels.py: pyrDown, Align2D), one tracked frame (tools/track_step.py: pyramid, single-pair
# alignment, warp prelude, Align2D, pose refinement, detector) and the batched pose refinement (tools/pose_opt_bench.py).
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-prof}"
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
CASE[poseopt]="$REPO/tools/pose_opt_bench.py 4096 200 nolatency"
for c in main solo n1000 n2000 kernels track poseopt; do
  mkdir -p "$OUT/$c"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$c/trace" -- python3 ${CASE[$c]} > "$OUT/$c/trace.log" 2>&1 || echo "trace $c failed"
  echo "traced $c"
done
for c in solo n1000 n2000 kernels track poseopt; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/$c/pmc_fetch" -- python3 ${CASE[$c]} > "$OUT/$c/pmc_fetch.log" 2>&1 || echo "pmc fetch $c failed"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/$c/pmc_write" -- python3 ${CASE[$c]} > "$OUT/$c/pmc_write.log" 2>&1 || echo "pmc write $c failed"
  echo "counted $c"
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/solo/pmc_sq" -- python3 ${CASE[solo]} > "$OUT/solo/pmc_sq.log" 2>&1 || echo "pmc sq failed"
# FP64 instruction mix of the alignment kernel (bench.py's fp64 block: flops per launch = 64 x (ADD + MUL + 2 FMA + TRANS))
timeout -k 5 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 --output-format csv -d "$OUT/solo/pmc_fp64" -- python3 ${CASE[solo]} > "$OUT/solo/pmc_fp64.log" 2>&1 || echo "pmc fp64 failed"
# keep the merge-back small: the per-dispatch CSVs of the long runs are summarised on the box
cd "$REPO"
python3 tools/summarize_profile.py "$OUT" > "$OUT/summary.txt" 2>&1
find "$OUT" -name "*_agent_info.csv" -delete
# the raw per-dispatch CSVs (tens of MB) stay on the box: the summaries above are what is kept (gpurun merges <= 64 MiB back)
for c in main solo n1000 n2000 kernels track poseopt; do rm -rf "$OUT/$c/trace" "$OUT/$c"/pmc_*/ ; done
echo profdone
