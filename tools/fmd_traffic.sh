#!/bin/bash
# HBM traffic of the fused FindMatchDirect kernel (FETCH_SIZE / WRITE_SIZE, separate --pmc passes), 64 frames x 800 candidates.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-fmdtraffic}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 $REPO/tools/fmd_bench.py 0 > "$OUT/$c.log" 2>&1 || echo "pass $c failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if ("match_kernel" in r["Kernel_Name"] or "warp_kernel" in r["Kernel_Name"] or "align2d" in r["Kernel_Name"]) and int(r["Grid_Size"]) > 100000:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
alg = 51200 * 105 + 2 * 64 * 640 * 480
for k, cs in acc.items():
    f, w = (sum(cs[c]) / max(len(cs[c]), 1) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    print(f"{k}: fetch {2.0 * f * 1024 / 1e6:.1f} MB (gfx950-corrected), write {w * 1024 / 1e6:.1f} MB per call = {(2.0 * f + w) * 1024 / alg:.2f} x {alg / 1e6:.1f} MB")
PY
rm -rf "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE
