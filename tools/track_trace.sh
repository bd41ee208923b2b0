#!/bin/bash
# One variant of the library under the one-call tracked frame: wall clock (pinned and pageable image) and the mean duration of
# every kernel of a frame from a rocprofv3 kernel trace. Used through tools/ab_lib.sh for same-box A/B:
#   tools/ab_lib.sh tools/track_trace.sh variants/lib_a.so variants/lib_b.so
label="${1:-current}"
REPO="$(cd "$(dirname "$0")/.." && pwd)"
export TMPDIR=/tmp
echo "== $label"
python3 "$REPO/tools/track_frame_bench.py" 900 pinned || exit 1
python3 "$REPO/tools/track_frame_bench.py" 900 || exit 1
rm -rf "/tmp/tt_$label"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "/tmp/tt_$label" -- python3 "$REPO/tools/track_frame_bench.py" 900 pinned > "/tmp/tt_$label.log" 2>&1 ) || { echo "trace failed"; tail -5 "/tmp/tt_$label.log"; exit 1; }
python3 - "/tmp/tt_$label" <<'PY'
import csv, glob, sys, collections
fs = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(fs[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
d = collections.OrderedDict()
for r in rows[len(rows) // 4:]:
    d.setdefault(r["Kernel_Name"].split("(")[0][:48], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in d.items():
    v.sort(); m = v[len(v) // 2]; tot += m
    print(f"   {k:48s} n {len(v):4d}  median {m:7.2f} us")
print(f"   sum of medians {tot:7.2f} us")
PY
