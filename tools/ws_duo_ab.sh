#!/bin/bash
# workspace kernel, 1025..2048 patches: one pair on one compute unit with its grid inputs in an HBM workspace
# (DSDTM_WS_NO_DUO=1) against one pair on two compute units, each half wholly in LDS; time and FETCH/WRITE traffic.
export DSDTM_PY_DIAG=1   # the DSDTM_* switches below exist in the diagnostic library only (build.py --diag)
REPO="$(cd "$(dirname "$0")/.." && pwd)"; cd "$REPO"
O=gpurun_out/${1:-wsduo}; mkdir -p $O
run() { timeout -k 10 120 python bench.py "$@" --steps 20 --warmup 3 --streams 1 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%s: kernel %.4f ms  %.0f /s' % ('$LABEL', d['roofline']['kernel_ms_avg'], d['value']))"; }
for rep in 1 2; do for nd in 1 0; do
  if [ $nd = 1 ]; then export DSDTM_WS_NO_DUO=1; else unset DSDTM_WS_NO_DUO; fi
  LABEL="no_duo $nd N=2000 1280x960 x256"; run --width 1280 --height 960 --patches 2000 --pairs 256 | tee -a $O/ab.txt
  LABEL="no_duo $nd N=1500 640x480 x1024"; run --patches 1500 | tee -a $O/ab.txt
  LABEL="no_duo $nd N=2000 1280x960 x256 streams4"; timeout -k 10 120 python bench.py --width 1280 --height 960 --patches 2000 --pairs 256 --steps 40 --warmup 4 --streams 4 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$LABEL: %.0f /s' % d['value'])" | tee -a $O/ab.txt
done; done
unset DSDTM_WS_NO_DUO
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d "$REPO/$O/pmc_duo_$c" -- python3 $REPO/bench.py --width 1280 --height 960 --patches 2000 --pairs 256 --steps 10 --warmup 2 --streams 1 --no-cpu --no-secondary > "$REPO/$O/pmc_duo_$c.log" 2>&1 || echo "pmc failed"
done
cd "$REPO"
python3 - "$O" <<'PY' | tee -a $O/ab.txt
import csv, glob, sys
o = sys.argv[1]; alg = 256 * 3378292
v = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    xs = [float(r["Counter_Value"]) for f in glob.glob(f"{o}/pmc_duo_{c}/**/*_counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))
          if "sparse_align" in r["Kernel_Name"] and r["Counter_Name"] == c]
    v[c] = sum(xs) / max(1, len(xs)) * 1024.0
tr = 2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]
print(f"traffic duo N=2000: FETCH_SIZE x2 {2 * v['FETCH_SIZE'] / 1e6:.0f} MB + WRITE_SIZE {v['WRITE_SIZE'] / 1e6:.0f} MB = {tr / 1e6:.0f} MB per launch = {tr / alg:.2f} x algorithmic ({alg / 1e6:.0f} MB)")
PY
rm -rf $O/pmc_*/
