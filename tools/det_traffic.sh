#!/bin/bash
# HBM traffic of the detector's batch passes (FETCH_SIZE / WRITE_SIZE, separate --pmc passes) for the library in place.
REPO="$(cd "$(dirname "$0")/.." && pwd)"
OUT="$REPO/gpurun_out/${1:-dettraffic}"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --pmc $c --output-format csv -d "$OUT/$c" -- python3 $REPO/tools/detect_bench.py 256 > "$OUT/$c.log" 2>&1 || echo "pass $c failed"
done
cd "$REPO"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "fast_" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = 0.0
for k, cs in acc.items():
    f, w = (sum(cs[c]) / max(len(cs[c]), 1) for c in ("FETCH_SIZE", "WRITE_SIZE"))
    b = 2.0 * f * 1024 + w * 1024; tot += b
    print(f"{k}: fetch {2.0 * f * 1024 / 1e6:.1f} MB (gfx950-corrected), write {w * 1024 / 1e6:.1f} MB per launch of 256 frames")
alg = 256 * 3 * (640 * 480 + 320 * 240 + 160 * 120 + 80 * 60 + 40 * 30)
print(f"total {tot / 1e6:.1f} MB = {tot / alg:.2f} x the algorithmic {alg / 1e6:.1f} MB")
PY
rm -rf "$OUT"/FETCH_SIZE "$OUT"/WRITE_SIZE
