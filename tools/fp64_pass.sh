#!/bin/bash
set -u
REPO="$(cd "$(dirname "$0")/.." && pwd)"; OUT="$REPO/gpurun_out/fp64"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 120 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU --output-format csv -d "$OUT/p" -- python3 $REPO/bench.py --steps 40 --warmup 5 --streams 1 --no-cpu --no-secondary > "$OUT/p.log" 2>&1 || echo failed
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sparse_align_reg_kernel" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:32s} {len(acc[k]):4d} {sum(acc[k])/len(acc[k]):.5g}")
PY
rm -rf "$OUT/p"
