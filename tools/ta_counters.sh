#!/bin/bash
# texture-address / L1 counters of the alignment kernel, one stream (diagnostic): tools/ta_counters.sh <outdir-under-gpurun_out>
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-ta}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="$REPO/bench.py --steps 60 --warmup 10 --streams 1 --no-cpu --no-secondary"
i=0
for grp in "TA_BUSY_avr TA_TA_BUSY_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" "GRBM_GUI_ACTIVE" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" \
           "TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  i=$((i+1))
  timeout -k 5 100 rocprofv3 --pmc $grp --output-format csv -d "$OUT/g$i" -- python3 $CMD > "$OUT/g$i.log" 2>&1 || echo "group $i ($grp) failed"
  echo "group $i done" | tee -a "$OUT/progress.txt"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sparse_align_reg_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:45s} dispatches {len(v):4d}  mean per dispatch {sum(v)/len(v):.4g}")
PY
