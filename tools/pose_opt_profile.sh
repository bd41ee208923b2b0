#!/bin/bash
# rocprofv3 passes behind profiles/r01_pose_opt_*: kernel trace + stats, then one counter group per pass
# (never --pmc together with other trace domains). Usage on the GPU box: tools/pose_opt_profile.sh <outdir-under-gpurun_out>
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-po_prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/tools/pose_opt_bench.py 16384 200 nolatency"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $B > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sq" -- python3 $B > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $B > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/pmc_lds" -- python3 $B > "$OUT/pmc_lds.log" 2>&1
echo profdone
