#!/bin/bash
# batches of 1024 pairs: register kernels (one slot per CU from 321 features) against the workspace kernel (2 x 4 waves)
export DSDTM_PY_DIAG=1   # the DSDTM_* switches below exist in the diagnostic library only (build.py --diag)
cd "$(dirname "$0")/.."
for n in 330 448 600 704; do for from in 704 320; do
  export DSDTM_WS_FROM=$from
  python bench.py --patches $n --steps 40 --warmup 4 --streams 4 --no-cpu --no-secondary 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('N=$n ws_from $from: %.0f /s (4 streams)' % d['value'])"
done; done
