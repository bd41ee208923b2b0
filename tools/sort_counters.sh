#!/bin/bash
# Round 5: why are the LATER iterations of the large-pair kernel slower when a pair's features are walked in image-row order?
# SQ / LDS / cache counters of the 1024 x 1000-patch launch with the device-side order on and off (separate --pmc passes).
export DSDTM_PY_DIAG=1   # the DSDTM_* switches below exist in the diagnostic library only (build.py --diag)
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-sortpmc}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$REPO/bench.py --patches 1000 --steps 10 --warmup 2 --streams 1 --preroll 0 --no-cpu --no-secondary"
for mode in sort nosort; do
  if [ $mode = nosort ]; then export DSDTM_WS_NO_SORT=1; else unset DSDTM_WS_NO_SORT; fi
  i=0
  for grp in "SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM" \
             "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum"; do
    i=$((i+1))
    timeout -k 5 240 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$mode/g$i" -- python3 $CMD > "$OUT/$mode.g$i.log" 2>&1 || echo "pass $mode g$i failed"
  done
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for mode in ("sort", "nosort"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(f"{out}/{mode}/g*/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "sparse_align_ws_kernel" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(mode, {k: round(v[0] / max(v[1], 1)) for k, v in sorted(acc.items())}, flush=True)
PY
find "$OUT" -name "*.csv" -size +1M -delete
echo done
