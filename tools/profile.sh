#!/bin/bash
# rocprofv3 passes behind profiles/*: kernel trace + stats, then one counter group per pass
# (never --pmc together with other trace domains). Usage on the GPU box: tools/profile.sh <outdir-under-gpurun_out>
set -e
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-prof}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py --steps 200 --warmup 20 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $B > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $B > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 $B > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d "$OUT/pmc_sq" -- python3 $B > "$OUT/pmc_sq.log" 2>&1
echo profdone
